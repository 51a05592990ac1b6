"""Reader / writer of MXNet's NDArray-list files (`prefix-%04d.params`) without importing mxnet.

What `mx.nd.load` / `mx.nd.save` do for the reference (lib/utils/load_model.py:10-30, save_model.py:10-24): a file is
  uint64 0x112 (list magic) | uint64 0 (reserved) | uint64 n | n x NDArray | uint64 n_names | n_names x (uint64 len, bytes)
and each dense NDArray is
  uint32 magic | [int32 stype (V2/V3 only)] | shape | int32 dev_type, int32 dev_id | int32 type_flag | raw little-endian data
with three on-disk generations of the shape record (all readable here, V2 written):
  legacy   (no magic: the first uint32 IS ndim)  ndim x uint32
  V1 0xF993FAC8                                   uint32 ndim, ndim x int64
  V2 0xF993FAC9 / V3 0xF993FACA (np shape)        int32 stype, then uint32/int32 ndim, ndim x int64
Sparse storage types (stype != 0) are rejected: DeepIM checkpoints hold dense float32 tensors only.

MXNet itself is not available in this environment (mxnet 1.x is a pip dependency of the reference, not vendored), so
this restates the published serialisation of src/ndarray/ndarray.cc; round trips and hand-built byte streams are tested in
tests/test_mx_params.py, files written by a real MXNet are not ("parity unpinned" for this format).
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC, V3_MAGIC = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
# mshadow type flags
_FLAG_TO_DTYPE = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64, 7: np.bool_}
_DTYPE_TO_FLAG = {np.dtype(v): k for k, v in _FLAG_TO_DTYPE.items()}


class _Reader(object):
    def __init__(self, buf):
        self.buf, self.pos = memoryview(buf), 0

    def read(self, fmt):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.buf):
            raise ValueError("truncated NDArray file (need {} bytes at offset {})".format(size, self.pos))
        out = struct.unpack_from("<" + fmt, self.buf, self.pos)
        self.pos += size
        return out if len(out) > 1 else out[0]

    def raw(self, n):
        if self.pos + n > len(self.buf):
            raise ValueError("truncated NDArray file (need {} data bytes at offset {})".format(n, self.pos))
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out


def _read_ndarray(r):
    magic = r.read("I")
    if magic in (V2_MAGIC, V3_MAGIC):
        stype = r.read("i")
        if stype != 0:
            raise ValueError("sparse NDArray (storage type {}) not supported".format(stype))
        ndim = r.read("i")
        if ndim < 0:  # V3 unknown shape
            return None
        shape = tuple(r.read("q") for _ in range(ndim))
        if magic == V2_MAGIC and ndim == 0:
            return None  # is_none()
    elif magic == V1_MAGIC:
        ndim = r.read("I")
        shape = tuple(r.read("q") for _ in range(ndim))
        if ndim == 0:
            return None
    else:  # legacy: the word just read is ndim, dims are uint32
        ndim = magic
        if ndim > 32:
            raise ValueError("not an NDArray record (bad magic 0x{:08x})".format(magic))
        shape = tuple(r.read("I") for _ in range(ndim))
        if ndim == 0:
            return None
    r.read("ii")  # context: dev_type, dev_id (arrays are always materialised on the host here)
    flag = r.read("i")
    if flag not in _FLAG_TO_DTYPE:
        raise ValueError("unknown mshadow type flag {}".format(flag))
    dt = np.dtype(_FLAG_TO_DTYPE[flag])
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    return np.frombuffer(r.raw(n * dt.itemsize), dtype=dt.newbyteorder("<")).reshape(shape).astype(dt, copy=True)


def nd_load(fname):
    """mx.nd.load: dict name -> numpy array (or a list when the file has no names)."""
    with open(fname, "rb") as f:
        r = _Reader(f.read())
    header, _ = r.read("QQ")
    if header != LIST_MAGIC:
        raise ValueError("{}: invalid NDArray file format (header 0x{:x})".format(fname, header))
    n = r.read("Q")
    arrays = [_read_ndarray(r) for _ in range(n)]
    n_names = r.read("Q")
    names = []
    for _ in range(n_names):
        ln = r.read("Q")
        names.append(bytes(r.raw(ln)).decode("utf-8"))
    if n_names == 0:
        return arrays
    if n_names != n:
        raise ValueError("{}: {} names for {} arrays".format(fname, n_names, n))
    return dict(zip(names, arrays))


def _write_ndarray(out, arr):
    arr = np.ascontiguousarray(arr)
    if arr.dtype not in _DTYPE_TO_FLAG:
        raise TypeError("dtype {} has no mshadow type flag".format(arr.dtype))
    out.append(struct.pack("<Ii", V2_MAGIC, 0))
    out.append(struct.pack("<I", arr.ndim))
    out.append(struct.pack("<{}q".format(arr.ndim), *arr.shape))
    if arr.ndim == 0:
        return
    out.append(struct.pack("<ii", 1, 0))  # Context: cpu(0)
    out.append(struct.pack("<i", _DTYPE_TO_FLAG[arr.dtype]))
    out.append(arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes())


def nd_save(fname, data):
    """mx.nd.save: `data` is a dict name -> array (written with names, in dict order) or a list of arrays."""
    names = list(data.keys()) if isinstance(data, dict) else []
    arrays = list(data.values()) if isinstance(data, dict) else list(data)
    out = [struct.pack("<QQQ", LIST_MAGIC, 0, len(arrays))]
    for a in arrays:
        _write_ndarray(out, a)
    out.append(struct.pack("<Q", len(names)))
    for nme in names:
        b = nme.encode("utf-8")
        out.append(struct.pack("<Q", len(b)))
        out.append(b)
    with open(fname, "wb") as f:
        f.write(b"".join(out))
