"""flax-0.3.0 checkpoint files <-> the engine's flat parameter / optimizer buffers.

What the reference writes (train_ncsn.py:395-399, read back at sample_ncsn.py:331-342 and :214-216,290-292):
``flax.training.checkpoints.save_checkpoint(dir, (optimizer, ema, early_stop), step, keep=N)`` -> file
``checkpoint_<step>`` holding ``flax.serialization.to_bytes`` of the tuple, i.e. msgpack of its state dict

    {'0': {'state':  {'step': i32[], 'param_states': <tree of {'grad_ema', 'grad_sq_ema'}>},      flax.optim.Adam
           'target': {'params': <tree>}},                                                         flax.nn.Model
     '1': {'mu': f, 'params': <tree>},                                           utils/train_utils.py:62-78 EMAHelper
     '2': {'min_delta', 'patience', 'best_metric', 'patience_count', 'should_stop'}}            :23-59 EarlyStopping

Arrays travel as msgpack ExtType(1, packb((shape, dtype.name, raw bytes))).  <tree> is the nested parameter dict of
the flax.nn module: children are auto-named ``<ClassName>_<n>`` in call order (models/ncsn.py:125-179,
models/shared.py:61-75), attention projections are explicitly named query / key / value / out with kernels
(E, H, d) / (H, d, E).

flax is not installable here, so both the wire format and the auto-naming rule are RESTATED (unpinned against flax itself).
The naming rule is the uncertain part, so the importer does not assume one: it generates the tree under every
plausible rule (one counter shared by all child modules, with or without the parameter-less ones taking a number;
one counter per class; attention registered as MultiHeadDotProductAttention or SelfAttention) and takes the rule
whose key sets match the file at every level.  The exporter's default is the shared counter with the attention module
registered as ``MultiHeadDotProductAttention_<n>``: in flax 0.3.0 ``nn.SelfAttention`` is
``MultiHeadDotProductAttention.partial(inputs_kv=None)`` and ``Module.partial`` keeps the parent's ``__name__``; the one
public artefact of that API we can cite from memory, the ViT checkpoints written with the same ``flax.nn`` calls
(``LayerNorm_0 / MultiHeadDotProductAttention_1 / LayerNorm_2 / MlpBlock_3`` inside each encoder block), shows both the
shared counter and that class name.
"""
from __future__ import annotations

import glob
import os
import re
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import msgpack
import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
NAMING_RULES = ("shared", "shared_params_only", "per_class")
ATTENTION_CLASS_NAMES = ("MultiHeadDotProductAttention", "SelfAttention")


# ------------------------------------------------------------------ msgpack wire format (flax.serialization)
def _ndarray_to_bytes(a: np.ndarray) -> bytes:
    a = np.asarray(a)
    if not a.flags.c_contiguous:                 # (np.ascontiguousarray would turn a 0-d array into shape (1,))
        a = a.copy(order="C")
    return msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True)


def _ndarray_from_bytes(data: bytes) -> np.ndarray:
    shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_to_bytes(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_to_bytes(np.asarray(x)))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
    raise TypeError(f"cannot serialise {type(x)}")


def _ext_unpack(code, data):
    if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
        return _ndarray_from_bytes(data)
    if code == _EXT_COMPLEX:
        re_, im_ = msgpack.unpackb(data)
        return complex(re_, im_)
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    """Later flax versions split arrays > 2^30 bytes into {'__msgpack_chunked_array__', 'shape', 'chunks'}."""
    if isinstance(tree, dict):
        if tree.get("__msgpack_chunked_array__"):
            chunks = tree["chunks"]
            flat = np.concatenate([np.asarray(chunks[str(i)]).ravel() for i in range(len(chunks))])
            return flat.reshape([int(s) for s in np.asarray(tree["shape"]).ravel()])
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def to_bytes(state_dict: Dict[str, Any]) -> bytes:
    return msgpack.packb(state_dict, default=_ext_pack, strict_types=True, use_bin_type=True)


def from_bytes(data: bytes) -> Dict[str, Any]:
    return _unchunk(msgpack.unpackb(data, ext_hook=_ext_unpack, raw=False, strict_map_key=False))


# ------------------------------------------------------------------ module tree of the reference networks
@dataclass
class Node:
    cls: str                                   # flax class name ('' for explicitly named children)
    name: Optional[str] = None                 # explicit name (attention projections)
    children: List["Node"] = field(default_factory=list)
    leaves: Dict[str, Tuple[str, Tuple[int, ...], Optional[Tuple[int, int]]]] = field(default_factory=dict)
    # leaves: flax leaf name -> (engine tensor name, flax shape, column range inside the engine tensor or None)

    @property
    def has_params(self) -> bool:
        return bool(self.leaves) or any(c.has_params for c in self.children)


def _dense(our: str, i: int, o: int, cols: Optional[Tuple[int, int]] = None, cls: str = "Dense", name=None,
           kshape=None, bshape=None) -> Node:
    return Node(cls, name, [], {"kernel": (our + ".kernel", kshape or (i, o), cols),
                                "bias": (our + ".bias", bshape or (o,), cols)})


def _ln(our: str, d: int) -> Node:
    return Node("LayerNorm", None, [], {"scale": (our + ".scale", (d,), None), "bias": (our + ".bias", (d,), None)})


def module_tree(cfg, attention_class: str = "MultiHeadDotProductAttention") -> Node:
    """Call-order tree of TransformerDDPM (models/ncsn.py:141-179) / DenseDDPM (:125-135)."""
    C, M = cfg.data_channels, cfg.mlp_dims
    E, F = getattr(cfg, "embed_channels", 128), getattr(cfg, "film_channels", 128)     # models/ncsn.py:151,173
    H = cfg.num_heads
    d = E // H

    def film(k: int) -> Node:                                   # models/ncsn.py:47-61
        p = f"film.{k}"
        return Node("DenseFiLM", None, [Node("NoiseEncoding"), _dense(p + ".fc1", F, 4 * F), _dense(p + ".fc2", 4 * F, 4 * F),
                                        _dense(p + ".ss", 4 * F, M, (0, M)), _dense(p + ".ss", 4 * F, M, (M, 2 * M))])

    def res(k: int) -> Node:                                    # models/shared.py:61-75 (no shortcut Dense: equal widths)
        p = f"res.{k}"
        return Node("DenseResBlock", None, [_ln(p + ".ln1", M), Node("FeaturewiseAffine"), _dense(p + ".fc1", M, M),
                                            _ln(p + ".ln2", M), Node("FeaturewiseAffine"), _dense(p + ".fc2", M, M)])

    top = Node(cfg.architecture)
    if cfg.architecture in ("TransformerDDPM", "TransformerDDPM4"):
        top.children += [Node("TransformerPositionalEncoding"), _dense("in_proj", C, E)]
        for l in range(cfg.num_layers):
            p = f"enc.{l}"
            attn = Node(attention_class, None, [
                _dense(p + ".attn.qkv", E, E, (0, E), "", "query", (E, H, d), (H, d)),
                _dense(p + ".attn.qkv", E, E, (E, 2 * E), "", "key", (E, H, d), (H, d)),
                _dense(p + ".attn.qkv", E, E, (2 * E, 3 * E), "", "value", (E, H, d), (H, d)),
                _dense(p + ".attn.out", E, E, None, "", "out", (H, d, E), (E,))])
            top.children += [_ln(p + ".ln1", E), attn, _ln(p + ".ln2", E), _dense(p + ".mlp.fc1", E, M),
                             _dense(p + ".mlp.fc2", M, E)]
        top.children += [_ln("ln_f", E), _dense("up", E, M)]
        for k in range(cfg.num_mlp_layers):
            top.children += [film(k), res(k)]
        top.children += [_ln("ln_o", M), _dense("out_proj", M, C)]
    elif cfg.architecture == "DenseDDPM":
        top.children.append(_dense("in_proj", C, M))
        for k in range(cfg.num_layers):
            top.children += [film(k), res(k)]
        top.children += [_ln("ln_o", M), _dense("out_proj", M, C)]
    else:
        raise ValueError(f"unsupported architecture {cfg.architecture}")
    return top


def _child_names(node: Node, rule: str) -> List[Optional[str]]:
    """Name of every child under ``rule`` (None for children that own no parameters and so never appear)."""
    names: List[Optional[str]] = []
    shared, per_class = 0, {}
    for c in node.children:
        if c.name is not None:
            names.append(c.name)
            continue
        if rule == "shared":
            n = shared
            shared += 1
        elif rule == "shared_params_only":
            n = shared
            shared += 1 if c.has_params else 0
        elif rule == "per_class":
            n = per_class.get(c.cls, 0)
            per_class[c.cls] = n + 1
        else:
            raise ValueError(f"unknown naming rule {rule!r}; one of {NAMING_RULES}")
        names.append(f"{c.cls}_{n}" if c.has_params else None)
    return names


def _walk(node: Node, rule: str, path: Tuple[str, ...] = ()):
    """Yield (flax path tuple, engine tensor name, flax shape, column range) of every leaf."""
    for leaf, (our, shape, cols) in node.leaves.items():
        yield path + (leaf,), our, shape, cols
    for c, nm in zip(node.children, _child_names(node, rule)):
        if nm is not None:
            yield from _walk(c, rule, path + (nm,))


def _get(tree, path):
    for k in path:
        tree = tree[k]
    return tree


def _set(tree, path, value):
    for k in path[:-1]:
        tree = tree.setdefault(k, {})
    tree[path[-1]] = value


def params_to_flax(named: Dict[str, np.ndarray], cfg, rule: str = "shared", attention_class: str = "MultiHeadDotProductAttention"):
    """Engine tensors (param_spec names / layouts) -> the nested flax parameter dict."""
    out: Dict[str, Any] = {}
    for path, our, shape, cols in _walk(module_tree(cfg, attention_class), rule):
        a = np.asarray(named[our], dtype=np.float32)
        if cols is not None:
            a = a[..., cols[0]:cols[1]]
        _set(out, path, np.ascontiguousarray(a.reshape(shape)))
    return out


def detect_naming(tree: Dict[str, Any], cfg) -> Tuple[str, str]:
    """The (rule, attention class name) under which ``tree``'s keys are exactly the module tree's."""
    shape_note = ""
    for ac in ATTENTION_CLASS_NAMES:
        for rule in NAMING_RULES:
            ok = True
            for path, _our, shape, _cols in _walk(module_tree(cfg, ac), rule):
                try:
                    leaf = _get(tree, path)
                except (KeyError, TypeError):
                    ok = False
                    break
                if tuple(np.shape(leaf)) != tuple(shape):      # same key, other tensor: not this rule (or not this model)
                    shape_note = shape_note or (f"; e.g. under rule {rule!r} {'/'.join(path)} has shape "
                                                f"{tuple(np.shape(leaf))}, expected {tuple(shape)}")
                    ok = False
                    break
            if ok:
                return rule, ac
    raise KeyError("parameter tree matches none of the known flax.nn naming rules; top-level keys: "
                   f"{sorted(tree)[:12]}{shape_note}")


def params_from_flax(tree: Dict[str, Any], cfg, template: Dict[str, Tuple[int, ...]]) -> Dict[str, np.ndarray]:
    """Nested flax parameter dict -> engine tensors.  ``template``: engine tensor name -> shape."""
    rule, ac = detect_naming(tree, cfg)
    out = {k: np.zeros(s, dtype=np.float32) for k, s in template.items()}
    seen = set()
    for path, our, _shape, cols in _walk(module_tree(cfg, ac), rule):
        a = np.asarray(_get(tree, path), dtype=np.float32)
        dst = out[our]
        if cols is None:
            dst[...] = a.reshape(dst.shape)
        else:
            dst[..., cols[0]:cols[1]] = a.reshape(dst.shape[:-1] + (cols[1] - cols[0],))
        seen.add(our)
    missing = set(template) - seen
    if missing:
        raise KeyError(f"module tree does not cover {sorted(missing)[:5]}")
    return out


# ------------------------------------------------------------------ whole checkpoints
def checkpoint_state_dict(cfg, params, grad_ema, grad_sq_ema, step: int, ema_params, ema_mu: float, early_stop: Dict[str, Any],
                          rule: str = "shared", attention_class: str = "MultiHeadDotProductAttention") -> Dict[str, Any]:
    """The state dict of (optimizer, ema, early_stop) as flax would build it; arguments are engine-named dicts."""
    f = lambda named: params_to_flax(named, cfg, rule, attention_class)
    m, v = f(grad_ema), f(grad_sq_ema)

    def zip_states(a, b):
        if isinstance(a, dict):
            return {k: zip_states(a[k], b[k]) for k in a}
        return {"grad_ema": a, "grad_sq_ema": b}

    es = dict(min_delta=0, patience=0, best_metric=float("inf"), patience_count=0, should_stop=False)
    es.update(early_stop or {})
    return {"0": {"state": {"step": np.asarray(step, dtype=np.int32), "param_states": zip_states(m, v)},
                  "target": {"params": f(params)}},
            "1": {"mu": float(ema_mu), "params": f(ema_params)},
            "2": es}


def split_state_dict(sd: Dict[str, Any], cfg, template: Dict[str, Tuple[int, ...]]):
    """Inverse of checkpoint_state_dict: (params, grad_ema, grad_sq_ema, step, ema_params, ema_mu, early_stop)."""
    opt, ema, es = sd["0"], sd["1"], sd["2"]
    ptree = opt["target"]["params"]
    params = params_from_flax(ptree, cfg, template)

    def unzip(t, which):
        if isinstance(t, dict) and set(t) == {"grad_ema", "grad_sq_ema"}:
            return t[which]
        return {k: unzip(v, which) for k, v in t.items()}

    ps = opt["state"]["param_states"]
    m = params_from_flax(unzip(ps, "grad_ema"), cfg, template)
    v = params_from_flax(unzip(ps, "grad_sq_ema"), cfg, template)
    step = int(np.asarray(opt["state"]["step"]).reshape(-1)[0])
    ema_params = params_from_flax(ema["params"], cfg, template)
    mu = float(np.asarray(ema["mu"]).reshape(-1)[0])
    scalar = lambda x: np.asarray(x).reshape(-1)[0].item() if isinstance(x, np.ndarray) else x
    return params, m, v, step, ema_params, mu, {k: scalar(x) for k, x in es.items()}


def is_flax_file(path: str) -> bool:
    """msgpack map header (fixmap / map16 / map32) vs safetensors (u64 JSON length, then '{')."""
    with open(path, "rb") as f:
        head = f.read(9)
    if len(head) == 9 and head[8:9] == b"{":
        return False
    return len(head) > 0 and (0x80 <= head[0] <= 0x8F or head[0] in (0xDE, 0xDF))


def write_file(path: str, state_dict: Dict[str, Any]) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(to_bytes(state_dict))
    os.replace(tmp, path)


def read_file(path: str) -> Dict[str, Any]:
    with open(path, "rb") as f:
        return from_bytes(f.read())
