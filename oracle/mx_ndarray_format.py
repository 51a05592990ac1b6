"""The on-disk format of `mx.nd.save` / `mx.nd.load`, written out a second time and independently of lib/utils/mx_params.py.
Oracle; test infrastructure only (tests/test_mx_params.py).

The reference keeps its checkpoints through MXNet (lib/utils/save_model.py:10-24 `mx.nd.save`, lib/utils/load_model.py:10-30
`mx.nd.load`); MXNet is a pip dependency of the reference (README.md:80-98: "tested under mxnet 1.2.0", source build pinned to
dmlc/mxnet@fc9e70b) and is NOT in /root/reference, nor importable in this image.  What follows restates the published serialisation
of that dependency -- `NDArray::Save` / `NDArray::Load` / `LegacyLoad` in src/ndarray/ndarray.cc, `TShape::Save` in
include/mxnet/tuple.h (nnvm/tuple.h in 1.2), `MXNDArraySave` in src/c_api/c_api.cc, dmlc-core's serializer for
`std::vector<std::string>` -- as a byte-by-byte builder and a byte-by-byte parser that share no code with the package's
struct-based ones.  No file written by a real MXNet exists here: "parity unpinned" for the format, as DESIGN.md section 4 says;
what this file buys is that the product's reader and writer are each checked against a second statement of the format
instead of against each other.

  file   = u64 0x112 | u64 0 | u64 n | n x record | u64 n_names | n_names x (u64 len, utf-8 bytes)      (c_api.cc, kMXAPINDArrayListMagic)
  record = legacy : u32 ndim, ndim x u32                                   | ctx | i32 type_flag | data   (before 0.12)
           V1     : u32 0xF993FAC8, u32 ndim, ndim x i64                   | ctx | i32 type_flag | data   (0.12 .. 1.0)
           V2     : u32 0xF993FAC9, i32 stype, u32 ndim, ndim x i64        | ctx | i32 type_flag | data   (1.0 .. 1.5; what 1.2.0 writes)
           V3     : u32 0xF993FACA, i32 stype, i32 ndim (-1 = unknown), .. | ctx | i32 type_flag | data   (numpy shape semantics)
  ctx    = i32 dev_type (1 cpu, 2 gpu, 3 cpu_pinned) | i32 dev_id; a record of ndim 0 (V1 / V2: "none") ends after the shape
  type_flag (mshadow/base.h): 0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64
  data is little-endian, C order."""
import numpy as np

FLAGS = {0: "<f4", 1: "<f8", 2: "<f2", 3: "u1", 4: "<i4", 5: "i1", 6: "<i8"}


def _u(v, n):
    return int(v).to_bytes(n, "little", signed=False)


def _i(v, n):
    return int(v).to_bytes(n, "little", signed=True)


def record(arr, generation="V2", dev_type=1, dev_id=0):
    """one dense NDArray as the named generation of MXNet wrote it"""
    arr = np.asarray(arr)
    flag = [k for k, v in FLAGS.items() if np.dtype(v) == arr.dtype.newbyteorder("<") or np.dtype(v) == arr.dtype][0]
    b = bytearray()
    if generation == "legacy":
        b += _u(arr.ndim, 4)
        for d in arr.shape:
            b += _u(d, 4)
    elif generation == "V1":
        b += _u(0xF993FAC8, 4) + _u(arr.ndim, 4)
        for d in arr.shape:
            b += _i(d, 8)
    elif generation == "V2":
        b += _u(0xF993FAC9, 4) + _i(0, 4) + _u(arr.ndim, 4)
        for d in arr.shape:
            b += _i(d, 8)
    elif generation == "V3":
        b += _u(0xF993FACA, 4) + _i(0, 4) + _i(arr.ndim, 4)
        for d in arr.shape:
            b += _i(d, 8)
    else:
        raise ValueError(generation)
    if arr.ndim == 0 and generation != "V3":
        return bytes(b)
    b += _i(dev_type, 4) + _i(dev_id, 4) + _i(flag, 4)
    b += np.ascontiguousarray(arr).astype(FLAGS[flag]).tobytes()
    return bytes(b)


def file_bytes(records, names):
    b = bytearray(_u(0x112, 8) + _u(0, 8) + _u(len(records), 8))
    for r in records:
        b += r
    b += _u(len(names), 8)
    for n in names:
        e = n.encode("utf-8")
        b += _u(len(e), 8) + e
    return bytes(b)


def parse(buf):
    """-> (arrays, names): the reader side, one cursor walking the bytes"""
    pos = [0]

    def take(n):
        s = buf[pos[0]:pos[0] + n]
        assert len(s) == n, "short file"
        pos[0] += n
        return s

    def u(n):
        return int.from_bytes(take(n), "little", signed=False)

    def i(n):
        return int.from_bytes(take(n), "little", signed=True)

    assert u(8) == 0x112
    u(8)
    arrays = []
    for _ in range(u(8)):
        head = u(4)
        if head in (0xF993FAC9, 0xF993FACA):
            assert i(4) == 0, "dense only"
            ndim = u(4) if head == 0xF993FAC9 else i(4)
            shape = [i(8) for _ in range(max(ndim, 0))]
            none = ndim == 0 if head == 0xF993FAC9 else ndim < 0
        elif head == 0xF993FAC8:
            ndim = u(4)
            shape = [i(8) for _ in range(ndim)]
            none = ndim == 0
        else:
            ndim = head
            shape = [u(4) for _ in range(ndim)]
            none = ndim == 0
        if none:
            arrays.append(None)
            continue
        i(4), i(4)
        dt = np.dtype(FLAGS[i(4)])
        count = 1
        for d in shape:
            count *= d
        arrays.append(np.frombuffer(take(count * dt.itemsize), dt).reshape(shape))
    names = [take(u(8)).decode("utf-8") for _ in range(u(8))]
    assert pos[0] == len(buf), "trailing bytes"
    return arrays, names
