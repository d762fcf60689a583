"""TFRecord files of tf.train.Example latents without TensorFlow.

The reference stores MusicVAE latents as TFRecords (scripts/transform_encoded_data.py:71-92 writes them,
utils/data_utils.py:44-60,159-190 reads them):

    Example.features.feature = {'inputs':      FloatList  (prod(shape) values),
                                'input_shape': Int64List  (len(shape) values)}

TFRecord framing (tensorflow/core/lib/io/record_writer): u64 length | u32 masked crc32c(length) | payload |
u32 masked crc32c(payload), masked = rotr15(crc) + 0xa282ead8.  The protobuf wire format is decoded by hand
(varints, length-delimited fields, packed and unpacked repeated scalars); tests/test_data_io.py cross-checks both
directions against messages built with the real protobuf runtime from a dynamically declared tf.train.Example.
"""
from __future__ import annotations

import glob
import os
import struct
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Union

import numpy as np

# ------------------------------------------------------------------ crc32c (Castagnoli), masked as TFRecord wants it
_POLY = 0x82F63B78
_TABLE = np.zeros(256, dtype=np.uint32)
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE[_i] = _c
_TABLE_L = [int(v) for v in _TABLE]


def crc32c(data: bytes) -> int:
    crc = 0xFFFFFFFF
    t = _TABLE_L
    for b in data:
        crc = t[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    crc = crc32c(data)
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------ record framing
def read_records(path: str, verify: Optional[str] = "length") -> Iterator[bytes]:
    """Payloads of one TFRecord file.  verify: None, 'length' (header CRC only, cheap) or 'full' (payload CRC too;
    pure-Python CRC, slow on 64 KiB latents)."""
    with open(os.path.expanduser(path), "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise IOError(f"{path}: truncated record header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify and masked_crc32c(head[:8]) != lcrc:
                raise IOError(f"{path}: corrupt record length")
            payload = f.read(length)
            tail = f.read(4)
            if len(payload) != length or len(tail) != 4:
                raise IOError(f"{path}: truncated record")
            if verify == "full" and masked_crc32c(payload) != struct.unpack("<I", tail)[0]:
                raise IOError(f"{path}: corrupt record payload")
            yield payload


def write_records(path: str, payloads: Iterable[bytes]) -> int:
    os.makedirs(os.path.dirname(os.path.abspath(os.path.expanduser(path))), exist_ok=True)
    n = 0
    with open(os.path.expanduser(path), "wb") as f:
        for p in payloads:
            head = struct.pack("<Q", len(p))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + p + struct.pack("<I", masked_crc32c(p)))
            n += 1
    return n


# ------------------------------------------------------------------ protobuf wire format, just enough for Example
def _varint(buf: bytes, pos: int):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field number, wire type, value) over one message; LEN values are memoryview slices."""
    pos, n = 0, len(buf)
    mv = memoryview(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(mv[pos:pos + 8]), pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = bytes(mv[pos:pos + ln]), pos + ln
        elif wt == 5:
            val, pos = bytes(mv[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, val


def _feature(buf: bytes):
    for fno, _wt, val in _fields(buf):
        if fno == 1:                                            # BytesList
            return [v for f2, _w, v in _fields(val) if f2 == 1]
        if fno == 2:                                            # FloatList: packed (LEN) or repeated fixed32
            parts = [np.frombuffer(v, dtype="<f4") for f2, _w, v in _fields(val) if f2 == 1]
            if not parts:
                return np.zeros((0,), np.float32)
            return np.concatenate(parts) if len(parts) != 1 else parts[0].copy()
        if fno == 3:                                            # Int64List: packed varints or repeated varint
            out: List[int] = []
            for f2, w, v in _fields(val):
                if f2 != 1:
                    continue
                if w == 0:
                    out.append(v)
                else:
                    pos = 0
                    while pos < len(v):
                        x, pos = _varint(v, pos)
                        out.append(x)
            return np.asarray([x - (1 << 64) if x >= (1 << 63) else x for x in out], dtype=np.int64)
    return np.zeros((0,), np.float32)                           # empty Feature


def parse_example(buf: bytes) -> Dict[str, Union[np.ndarray, List[bytes]]]:
    """tf.train.Example -> {feature name: float32 array | int64 array | list of bytes}."""
    out: Dict[str, Union[np.ndarray, List[bytes]]] = {}
    for fno, _wt, features in _fields(buf):
        if fno != 1:
            continue
        for f2, _w, entry in _fields(features):
            if f2 != 1:
                continue
            key, feat = None, b""
            for f3, _w3, v in _fields(entry):
                if f3 == 1:
                    key = v.decode("utf-8")
                elif f3 == 2:
                    feat = v
            if key is not None:
                out[key] = _feature(feat)
    return out


def _enc_varint(x: int) -> bytes:
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _len_field(fno: int, payload: bytes) -> bytes:
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(features: Dict[str, Union[np.ndarray, Sequence[bytes]]]) -> bytes:
    """The inverse of parse_example (packed repeated scalars, map entries in sorted key order like protobuf's
    deterministic serialisation)."""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, np.ndarray) and v.dtype.kind == "f":
            feat = _len_field(2, _len_field(1, np.ascontiguousarray(v, dtype="<f4").tobytes()) if v.size else b"")
        elif isinstance(v, np.ndarray) and v.dtype.kind in "iu":
            feat = _len_field(3, _len_field(1, b"".join(_enc_varint(int(x)) for x in v.ravel())) if v.size else b"")
        else:
            feat = _len_field(1, b"".join(_len_field(1, bytes(b)) for b in v))
        entries += _len_field(1, _len_field(1, key.encode("utf-8")) + _len_field(2, feat))
    return _len_field(1, entries)


# ------------------------------------------------------------------ the reference's latent datasets
def latent_example(x: np.ndarray) -> bytes:
    """scripts/transform_encoded_data.py:71-92 ('flatten' / default mode): inputs + input_shape."""
    x = np.asarray(x, dtype=np.float32)
    return make_example({"inputs": x.reshape(-1), "input_shape": np.asarray(x.shape, dtype=np.int64)})


def write_latents(path: str, arrays: Iterable[np.ndarray]) -> int:
    return write_records(path, (latent_example(a) for a in arrays))


def read_latents(file_pattern: str, shape: Sequence[int], limit: Optional[int] = None, verify: Optional[str] = "length") -> np.ndarray:
    """All examples of the files matching ``file_pattern`` (sorted), each reshaped to its stored input_shape and
    checked against ``shape`` (utils/data_utils.py:44-60: FixedLenFeature([prod(shape)]) + reshape(input_shape))."""
    files = sorted(glob.glob(os.path.expanduser(file_pattern)))
    if not files:
        raise FileNotFoundError(f"no TFRecord files match {file_pattern}")
    want = int(np.prod(shape))
    out: List[np.ndarray] = []
    for fn in files:
        for rec in read_records(fn, verify):
            ex = parse_example(rec)
            x, s = ex.get("inputs"), ex.get("input_shape")
            if not isinstance(x, np.ndarray) or x.dtype != np.float32 or x.size != want:
                raise ValueError(f"{fn}: 'inputs' holds {getattr(x, 'size', None)} floats, expected {want} for shape {tuple(shape)}")
            if s is None or len(s) != len(shape):
                raise ValueError(f"{fn}: 'input_shape' = {s}, expected {len(shape)} dims")
            out.append(x.reshape([int(v) for v in s]))
            if limit is not None and len(out) >= limit:
                return np.stack(out)
    return np.stack(out) if out else np.zeros((0, *shape), np.float32)
