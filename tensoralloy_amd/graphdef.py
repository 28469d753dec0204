"""
Reader for the reference's frozen TensorFlow `GraphDef` model files (`*.pb`), without TensorFlow.

`BasicNN.export` (reference nn/basic.py:1075-1092) freezes its graph with the variables turned into
`Const` nodes and bakes `Transformer/params` (JSON) and `Metadata/*` string constants into it;
`TensorAlloyCalculator.__init__` (calculator.py:128-170) reads them back through TF. Everything this
package needs from such a file is in `Const` nodes: the JSON / metadata strings, the MLP weights
`Atomic/<El>/Conv1d{j}/{kernel,bias}`, `Atomic/<El>/Output/{kernel,bias}`, the min-max bounds and
the constants of empirical EAM potentials `EAM/Shared/<El>/<param>`. The protobuf wire format is
self-describing (varint / 64-bit / length-delimited / 32-bit fields), so a 100-line parser that
knows five message layouts (GraphDef, NodeDef, AttrValue, TensorProto, TensorShapeProto:
tensorflow/core/framework/*.proto) is enough; the graph's ops are never executed.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, Tuple

import numpy as np


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message; length-delimited values are `bytes`."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        if pos > n:
            raise ValueError("truncated protobuf message")
        yield field, wire, val


# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
           9: np.int64, 10: np.bool_}
_DT_STRING = 7


def _signed(v: int, bits: int = 64) -> int:
    return v - (1 << bits) if v >> (bits - 1) else v


def _shape(buf: bytes):
    dims = []
    for field, _, val in _fields(buf):
        if field == 2:  # Dim
            size = 0
            for f2, _, v2 in _fields(val):
                if f2 == 1:
                    size = _signed(v2)
            dims.append(size)
    return tuple(dims)


def _tensor(buf: bytes):
    """TensorProto -> numpy array (numeric dtypes) or list of `bytes` (DT_STRING)."""
    dtype, shape, content = 0, (), None
    floats, doubles, ints, int64s, strings, bools = [], [], [], [], [], []
    for field, wire, val in _fields(buf):
        if field == 1:
            dtype = val
        elif field == 2:
            shape = _shape(val)
        elif field == 4:
            content = val
        elif field == 5:      # float_val, packed or not
            floats += list(struct.unpack(f"<{len(val) // 4}f", val)) if wire == 2 else [struct.unpack("<f", val)[0]]
        elif field == 6:
            doubles += list(struct.unpack(f"<{len(val) // 8}d", val)) if wire == 2 else [struct.unpack("<d", val)[0]]
        elif field in (7, 10, 11):   # int_val / int64_val / bool_val
            dst = ints if field == 7 else (int64s if field == 10 else bools)
            if wire == 2:
                p = 0
                while p < len(val):
                    v, p = _varint(val, p)
                    dst.append(_signed(v))
            else:
                dst.append(_signed(val))
        elif field == 8:
            strings.append(val)
    if dtype == _DT_STRING:
        return strings
    if dtype not in _DTYPES:
        raise ValueError(f"unsupported tensor dtype {dtype}")
    np_dtype = _DTYPES[dtype]
    n = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        arr = np.frombuffer(content, dtype=np.dtype(np_dtype).newbyteorder("<")).astype(np_dtype)
    else:
        vals = {np.float32: floats, np.float64: doubles, np.int64: int64s, np.bool_: bools}.get(np_dtype, ints)
        arr = np.array(vals, dtype=np_dtype)
        if arr.size == 1 and n > 1:   # a single value stands for the whole tensor
            arr = np.full(n, arr[0], dtype=np_dtype)
        elif arr.size == 0:
            arr = np.zeros(n, dtype=np_dtype)
    return arr.reshape(shape) if shape else (arr.reshape(()) if arr.size == 1 else arr)


def _open(path: str) -> bytes:
    import gzip
    with open(path, "rb") as fp:
        buf = fp.read()
    return gzip.decompress(buf) if buf[:2] == b"\x1f\x8b" else buf


def read_node_ops(path: str) -> Dict[str, str]:
    """{node name: op type} of every node (which activation a frozen AtomicNN graph applies is only
    visible in its op types)."""
    out = {}
    for field, wire, node in _fields(_open(path)):
        if field != 1 or wire != 2:
            continue
        name = op = None
        for f2, _, v2 in _fields(node):
            if f2 == 1:
                name = v2.decode("utf-8", "replace")
            elif f2 == 2:
                op = v2.decode("utf-8", "replace")
        if name is not None:
            out[name] = op
    return out


def read_constants(path: str) -> Dict[str, object]:
    """{node name: value} for every `Const` node of a frozen GraphDef file (plain or gzip'ed)."""
    buf = _open(path)
    out = {}
    for field, wire, node in _fields(buf):
        if field != 1 or wire != 2:   # GraphDef.node
            continue
        name, op, value = None, None, None
        for f2, w2, v2 in _fields(node):
            if f2 == 1:
                name = v2.decode("utf-8", "replace")
            elif f2 == 2:
                op = v2.decode("utf-8", "replace")
            elif f2 == 5 and w2 == 2:  # attr map entry {key = 1, value = 2}
                key, attr = None, None
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        key = v3
                    elif f3 == 2:
                        attr = v3
                if key == b"value" and attr is not None:
                    for f4, w4, v4 in _fields(attr):
                        if f4 == 8 and w4 == 2:   # AttrValue.tensor
                            value = v4
        if op == "Const" and name is not None and value is not None:
            try:
                out[name] = _tensor(value)
            except ValueError as exc:
                # a constant this reader cannot decode (an encoding it does not know): harmless for
                # the graph's own bookkeeping nodes, but never for a constant of the MODEL -- dropping
                # one silently would load a model that runs and returns wrong energies
                if name.startswith(("Atomic/", "EAM/", "ADP/", "Transformer/", "Metadata/")):
                    raise ValueError(f"{path}: constant '{name}' cannot be decoded: {exc}") from exc
                continue
    return out


def _text(value) -> str:
    if isinstance(value, list):
        value = value[0] if value else b""
    if isinstance(value, bytes):
        return value.decode("utf-8")
    return str(value)


def read_graph_model(path: str) -> Tuple[dict, Dict[str, object]]:
    """(metadata, constants) of a model exported by the reference: `metadata` holds the decoded
    `Transformer/params` and `Metadata/*` strings under the names the reference uses
    (calculator.py:128-170), `constants` every other `Const` node."""
    consts = read_constants(path)
    if "Transformer/params" not in consts:
        raise ValueError(f"{path}: no Transformer/params constant: not a model exported by "
                         f"tensoralloy (BasicNN.export, basic.py:1075-1092)")
    meta = {"Transformer/params": _text(consts["Transformer/params"])}
    for name, value in consts.items():
        if name.startswith("Metadata/"):
            meta[name] = _text(value)
    return meta, consts
