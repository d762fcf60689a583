"""CPU tests of the TFRecord / tf.train.Example reader-writer and of the dataset opening order
(input_pipeline.py:113-235).  The hand-written protobuf wire code is cross-checked in both directions against the
real protobuf runtime with a dynamically declared tf.train.Example (tensorflow/core/example/{example,feature}.proto)."""
import os
import struct

import numpy as np
import pytest

import ddpm_oracle as O


def _tf_example_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="smd_test_example.proto", package="smdtest", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def fld(m, name, number, ftype, label=T.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
        f = m.field.add(name=name, number=number, type=ftype, label=label)
        if type_name:
            f.type_name = type_name
        if packed is not None:
            f.options.packed = packed
        if oneof is not None:
            f.oneof_index = oneof
        return f

    fld(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    fld(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    fld(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    feat = msg("Feature")
    feat.oneof_decl.add(name="kind")
    fld(feat, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".smdtest.BytesList", oneof=0)
    fld(feat, "float_list", 2, T.TYPE_MESSAGE, type_name=".smdtest.FloatList", oneof=0)
    fld(feat, "int64_list", 3, T.TYPE_MESSAGE, type_name=".smdtest.Int64List", oneof=0)
    feats = msg("Features")
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    fld(entry, "key", 1, T.TYPE_STRING)
    fld(entry, "value", 2, T.TYPE_MESSAGE, type_name=".smdtest.Feature")
    fld(feats, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".smdtest.Features.FeatureEntry")
    fld(msg("Example"), "features", 1, T.TYPE_MESSAGE, type_name=".smdtest.Features")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("smdtest.Example"))


def test_crc32c_and_record_framing(tmp_path):
    import smd_amd.tfrecord as T
    assert T.crc32c(b"123456789") == 0xE3069283                  # the standard CRC-32C check value
    assert T.crc32c(b"") == 0 and T.crc32c(bytes(32)) == 0x8A9136AA      # RFC 3720 B.4: 32 zero bytes
    payloads = [b"", b"a", os.urandom(1000)]
    p = str(tmp_path / "x.tfrecord")
    assert T.write_records(p, payloads) == 3
    assert list(T.read_records(p, "full")) == payloads
    raw = open(p, "rb").read()
    assert struct.unpack("<Q", raw[:8])[0] == 0 and len(raw) == sum(16 + len(x) for x in payloads)
    bad = bytearray(raw)
    bad[-5] ^= 1                                                  # flip a payload bit of the last record
    open(p, "wb").write(bytes(bad))
    assert len(list(T.read_records(p, "length"))) == 3
    with pytest.raises(IOError):
        list(T.read_records(p, "full"))
    open(p, "wb").write(raw[:-3])
    with pytest.raises(IOError):
        list(T.read_records(p))


def test_example_wire_format_against_protobuf_runtime():
    import smd_amd.tfrecord as T
    Example = _tf_example_classes()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((32, 512)).astype(np.float32)
    # protobuf runtime -> our parser
    ex = Example()
    ex.features.feature["inputs"].float_list.value.extend(x.reshape(-1).tolist())
    ex.features.feature["input_shape"].int64_list.value.extend([32, 512])
    ex.features.feature["neg"].int64_list.value.extend([-1, -(1 << 40), 7])
    ex.features.feature["blob"].bytes_list.value.extend([b"ab", b""])
    got = T.parse_example(ex.SerializeToString())
    assert np.array_equal(got["inputs"], x.reshape(-1)) and got["inputs"].dtype == np.float32
    assert got["input_shape"].tolist() == [32, 512] and got["neg"].tolist() == [-1, -(1 << 40), 7]
    assert got["blob"] == [b"ab", b""]
    # our writer -> protobuf runtime
    back = Example()
    back.ParseFromString(T.make_example({"inputs": x.reshape(-1), "input_shape": np.asarray([32, 512], np.int64),
                                         "neg": np.asarray([-5], np.int64), "blob": [b"xyz"]}))
    f = back.features.feature
    assert np.array_equal(np.asarray(f["inputs"].float_list.value, np.float32), x.reshape(-1))
    assert list(f["input_shape"].int64_list.value) == [32, 512] and list(f["neg"].int64_list.value) == [-5]
    assert list(f["blob"].bytes_list.value) == [b"xyz"]
    # unpacked repeated floats (proto2-style writers) are accepted too
    unpacked = T._len_field(1, T._len_field(1, T._len_field(1, b"v") + T._len_field(2, T._len_field(
        2, b"".join(b"\x0d" + struct.pack("<f", v) for v in (1.5, -2.0))))))
    assert T.parse_example(unpacked)["v"].tolist() == [1.5, -2.0]


def test_open_dataset_from_tfrecords_follows_the_reference_order(tmp_path):
    """slice -> batch(drop_remainder) -> per-split min/max (+cache pickles) -> per-split normalisation."""
    import smd_amd.data as D
    import smd_amd.tfrecord as T
    rng = np.random.default_rng(1)
    train = rng.standard_normal((21, 4, 16)).astype(np.float32) * 3
    evalx = rng.standard_normal((10, 4, 16)).astype(np.float32)
    T.write_latents(str(tmp_path / "train-00000-of-00002.tfrecord"), train[:11])
    T.write_latents(str(tmp_path / "train-00001-of-00002.tfrecord"), train[11:])
    T.write_latents(str(tmp_path / "eval-00000-of-00001.tfrecord"), evalx)
    assert np.array_equal(T.read_latents(str(tmp_path / "train-*.tfrecord"), (4, 16)), train)
    with pytest.raises(ValueError):
        T.read_latents(str(tmp_path / "train-*.tfrecord"), (4, 8))
    slice_idx = np.asarray([1, 5, 6, 15])
    tr, ev = D.open_dataset(str(tmp_path), 4, (4, 4), None, 0, 1, True, slice_idx, None, data_shape=(4, 16),
                            slice_ckpt="/x/mel_slice.pkl")
    kept_t, kept_e = train[:20][..., slice_idx], evalx[:8][..., slice_idx]       # remainders dropped BEFORE min/max
    assert tr.examples == 5 and ev.examples == 2
    assert tr.min == pytest.approx(float(kept_t.min())) and tr.max == pytest.approx(float(kept_t.max()))
    assert ev.min == pytest.approx(float(kept_e.min())) and ev.max == pytest.approx(float(kept_e.max()))   # its own range
    want = O.normalize_dataset(kept_t, np.float32(kept_t.min()), np.float32(kept_t.max()))
    got = np.concatenate([b.numpy() for b in tr])                                # the training split: seeded reshuffle
    order = tr.epoch_order(0).numpy()
    assert sorted(order.tolist()) == list(range(20)) and np.allclose(got, want[order], atol=1e-6)
    plain, _ = D.open_dataset(str(tmp_path), 4, (4, 4), None, 0, 1, True, slice_idx, None, data_shape=(4, 16),
                              slice_ckpt="/x/mel_slice.pkl", shuffle=False)
    assert np.allclose(np.concatenate([b.numpy() for b in plain]), want, atol=1e-6)
    assert float(tr.array.min()) == -1.0 and float(tr.array.max()) == 1.0
    # the reference's cache files ({dataset}/cache/{split}_{config}_{min,max}.pkl) exist and win on the next open
    for split in ("train", "eval"):
        for which in ("min", "max"):
            assert os.path.exists(tmp_path / "cache" / f"{split}_mel_slice_{which}.pkl")
    D.save(np.float32(-100.0), str(tmp_path / "cache" / "train_mel_slice_min.pkl"))
    tr2, _ = D.open_dataset(str(tmp_path), 4, (4, 4), None, 0, 1, True, slice_idx, None, data_shape=(4, 16),
                            slice_ckpt="/x/mel_slice.pkl")
    assert tr2.min == -100.0
    # inverse transform puts the slice back into 512-wide rows (random fill elsewhere, float64) -- :78-110
    inv = D.inverse_data_transform(np.concatenate([b.numpy() for b in plain]), True, None, tr.min, tr.max, slice_idx, None,
                                   out_channels=16)
    assert inv.dtype == np.float64 and inv.shape == (20, 4, 16)
    assert np.allclose(inv[..., slice_idx], kept_t, atol=1e-5)


# ---------------------------------------------------------------------------------------------- property tests
from hypothesis import given, settings, strategies as st_  # noqa: E402
import hypothesis.extra.numpy as hnp  # noqa: E402

_names = st_.text(alphabet="abcdefghijklmnopqrstuvwxyz_0123456789", min_size=1, max_size=12)
_floats = hnp.arrays(np.float32, st_.integers(0, 40), elements=st_.floats(-1e6, 1e6, width=32))
_ints = hnp.arrays(np.int64, st_.integers(0, 40), elements=st_.integers(-(2 ** 62), 2 ** 62))
_bytes = st_.lists(st_.binary(max_size=20), max_size=4)


@settings(max_examples=60, deadline=None)
@given(st_.dictionaries(_names, st_.one_of(_floats, _ints, _bytes), max_size=5))
def test_example_round_trip_and_protobuf_agreement_property(features):
    """Any Example our writer produces is read back identically by our parser AND by the protobuf runtime."""
    import smd_amd.tfrecord as T
    raw = T.make_example(features)
    got = T.parse_example(raw)
    ex = _tf_example_classes()()
    ex.ParseFromString(raw)
    assert set(got) == set(features) == set(ex.features.feature)
    for k, v in features.items():
        f = ex.features.feature[k]
        if isinstance(v, np.ndarray) and v.dtype == np.float32:
            assert np.array_equal(got[k], v) and np.array_equal(np.asarray(f.float_list.value, np.float32), v)
        elif isinstance(v, np.ndarray):
            assert np.array_equal(got[k], v) and list(f.int64_list.value) == v.tolist()
        else:
            assert list(got[k]) == list(v) if len(v) else len(got[k]) == 0
            assert list(f.bytes_list.value) == list(v)
    # and the runtime's own serialisation of the same message parses to the same features
    again = T.parse_example(ex.SerializeToString())
    for k, v in features.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(again[k], v)


_dtypes = st_.sampled_from([np.float32, np.float64, np.int32, np.int64, np.uint8, np.bool_, np.float16])


@settings(max_examples=60, deadline=None)
@given(st_.recursive(
    st_.one_of(st_.integers(-2 ** 31, 2 ** 31), st_.floats(allow_nan=False), st_.booleans(),
               _dtypes.flatmap(lambda d: hnp.arrays(d, hnp.array_shapes(min_dims=0, max_dims=3, max_side=5)))),
    lambda children: st_.dictionaries(_names, children, max_size=4), max_leaves=12))
def test_flax_msgpack_round_trip_property(tree):
    import smd_amd.flax_io as FI

    def same(a, b):
        if isinstance(a, dict):
            return isinstance(b, dict) and set(a) == set(b) and all(same(a[k], b[k]) for k in a)
        if isinstance(a, np.ndarray):
            return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
        return a == b and type(a) is type(b)

    if not isinstance(tree, dict):
        tree = {"leaf": tree}
    assert same(tree, FI.from_bytes(FI.to_bytes(tree)))
