"""Writes tests/golden/flax_dense_ddpm_tiny.msgpack: the bytes flax-0.3.0's
``checkpoints.save_checkpoint(dir, (optimizer, ema, early_stop), step)`` would put on disk for a tiny DenseDDPM
(/root/reference/train_ncsn.py:395-399, read back at sample_ncsn.py:331-342), assembled BY HAND from the msgpack
specification and the flax.serialization layout -- no msgpack library, none of smd_amd's code, no model code:

  * file        = msgpack map of the state dict of the tuple: keys '0' (flax.optim.Optimizer), '1' (EMAHelper), '2' (EarlyStopping)
  * ndarray     = ext type 1 whose payload is the msgpack array [shape list, dtype name, raw C-order bytes (bin)]
  * python int / float / bool stay native msgpack scalars; dict keys are strings, emitted in sorted order
  * module tree = flax.nn auto-names ``<Class>_<n>``, n = position among the parent's submodules in call order
                  (parameter-less NoiseEncoding / FeaturewiseAffine take a number and leave no entry)

Model: DenseDDPM (models/ncsn.py:125-135) with data_channels 4, mlp_dims 8, num_layers 1 and a FiLM embedding of 4 channels
(the reference hard-codes 128; the fixture shrinks it so the file stays a few KB -- the tree shape is unchanged).

Every leaf holds  value[i] = leaf_number + i / 1024  (leaf_number = position in LEAVES below), so a reader can be checked
without this script: parameters = value, Adam grad_ema = -value, grad_sq_ema = 2 * value, EMA parameters = value + 100.

Run:  python tests/golden/make_flax_fixture.py
"""
import os
import struct

import numpy as np

C, M, F = 4, 8, 4

# (path in the flax parameter tree, shape) in call order -- written out literally, one line per tensor
LEAVES = [
    (("Dense_0", "kernel"), (C, M)), (("Dense_0", "bias"), (M,)),                                   # nn.Dense(x, mlp_dims)       :129
    # DenseFiLM_1: NoiseEncoding_0 has no parameters                                                                             :47-61
    (("DenseFiLM_1", "Dense_1", "kernel"), (F, 4 * F)), (("DenseFiLM_1", "Dense_1", "bias"), (4 * F,)),
    (("DenseFiLM_1", "Dense_2", "kernel"), (4 * F, 4 * F)), (("DenseFiLM_1", "Dense_2", "bias"), (4 * F,)),
    (("DenseFiLM_1", "Dense_3", "kernel"), (4 * F, M)), (("DenseFiLM_1", "Dense_3", "bias"), (M,)),        # scale
    (("DenseFiLM_1", "Dense_4", "kernel"), (4 * F, M)), (("DenseFiLM_1", "Dense_4", "bias"), (M,)),        # shift
    # DenseResBlock_2: FeaturewiseAffine_1 / _4 have no parameters                                              models/shared.py:61-75
    (("DenseResBlock_2", "LayerNorm_0", "scale"), (M,)), (("DenseResBlock_2", "LayerNorm_0", "bias"), (M,)),
    (("DenseResBlock_2", "Dense_2", "kernel"), (M, M)), (("DenseResBlock_2", "Dense_2", "bias"), (M,)),
    (("DenseResBlock_2", "LayerNorm_3", "scale"), (M,)), (("DenseResBlock_2", "LayerNorm_3", "bias"), (M,)),
    (("DenseResBlock_2", "Dense_5", "kernel"), (M, M)), (("DenseResBlock_2", "Dense_5", "bias"), (M,)),
    (("LayerNorm_3", "scale"), (M,)), (("LayerNorm_3", "bias"), (M,)),                              # nn.LayerNorm(x)             :133
    (("Dense_4", "kernel"), (M, C)), (("Dense_4", "bias"), (C,)),                                   # nn.Dense(x, z_dims)         :134
]
STEP, MU = 4321, 0.999
EARLY_STOP = dict(min_delta=0, patience=1, best_metric=0.0625, patience_count=1, should_stop=False)


# ---------------------------------------------------------------- msgpack, by the spec
def m_str(s):
    b = s.encode()
    if len(b) < 32:
        return bytes([0xA0 | len(b)]) + b
    assert len(b) < 256
    return b"\xd9" + bytes([len(b)]) + b


def m_int(v):
    if 0 <= v < 128:
        return bytes([v])
    if 0 <= v < 65536:
        return b"\xcd" + struct.pack(">H", v)
    return b"\xd2" + struct.pack(">i", v)


def m_bin(b):
    if len(b) < 256:
        return b"\xc4" + bytes([len(b)]) + b
    if len(b) < 65536:
        return b"\xc5" + struct.pack(">H", len(b)) + b
    return b"\xc6" + struct.pack(">I", len(b)) + b


def m_ext(code, b):
    if len(b) < 256:
        return b"\xc7" + bytes([len(b), code]) + b
    if len(b) < 65536:
        return b"\xc8" + struct.pack(">H", len(b)) + bytes([code]) + b
    return b"\xc9" + struct.pack(">I", len(b)) + bytes([code]) + b


def m_ndarray(a):
    shape = bytes([0x90 | a.ndim]) + b"".join(m_int(int(d)) for d in a.shape)          # fixarray of ints
    payload = bytes([0x93]) + shape + m_str(a.dtype.name) + m_bin(a.tobytes(order="C"))  # [shape, dtype, bytes]
    return m_ext(1, payload)


def m_value(v):
    if isinstance(v, dict):
        assert len(v) < 16
        return bytes([0x80 | len(v)]) + b"".join(m_str(k) + m_value(v[k]) for k in sorted(v))
    if isinstance(v, np.ndarray):
        return m_ndarray(v)
    if isinstance(v, bool):
        return b"\xc3" if v else b"\xc2"
    if isinstance(v, int):
        return m_int(v)
    if isinstance(v, float):
        return b"\xcb" + struct.pack(">d", v)
    raise TypeError(type(v))


def leaf_value(n, shape):
    return (np.float32(n) + np.arange(int(np.prod(shape)), dtype=np.float32) / np.float32(1024)).reshape(shape)


def tree(fn):
    out = {}
    for n, (path, shape) in enumerate(LEAVES):
        d = out
        for k in path[:-1]:
            d = d.setdefault(k, {})
        d[path[-1]] = fn(leaf_value(n, shape))
    return out


def main():
    state = {
        "0": {"state": {"step": np.asarray(STEP, dtype=np.int32),
                        "param_states": tree(lambda v: {"grad_ema": -v, "grad_sq_ema": 2 * v})},
              "target": {"params": tree(lambda v: v)}},
        "1": {"mu": MU, "params": tree(lambda v: v + np.float32(100))},
        "2": dict(EARLY_STOP),
    }
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flax_dense_ddpm_tiny.msgpack")
    with open(out, "wb") as f:
        f.write(m_value(state))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
