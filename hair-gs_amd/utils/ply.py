"""Minimal PLY reader / writer (numpy only) for the files the reference exchanges through `plyfile`:
Gaussian clouds (scene/gaussian_model.py:283-412), the five-element strand model (scene/hair_gaussian_model.py:310-466)
and COLMAP point clouds (data/dataset_readers.py:181-213).

Elements are (name, structured array) pairs with scalar properties only, which is all those files contain.  Files are
written as `binary_little_endian 1.0` with plyfile's type names (`float`, `int`, `uchar`, ...), i.e. what
`PlyData([...]).write(path)` produces on a little-endian host; binary (either endianness) and ASCII files are read.
SURVEY.md 8f n4: the formats are restated from the cited reference lines; plyfile is not in the image, so no file was
produced by the reference itself ("parity unpinned" for the byte layout; the layout is the public PLY specification)."""
import numpy as np

# PLY scalar type names (both spellings) <-> numpy
_PLY_TO_NP = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
              "f8": "double"}


def write_ply(path, elements):
    """elements: iterable of (name, structured ndarray).  One `element` block per pair, in order."""
    header = ["ply", "format binary_little_endian 1.0"]
    blobs = []
    for name, arr in elements:
        arr = np.asarray(arr)
        if arr.dtype.names is None:
            raise TypeError(f"element {name!r}: structured array expected")
        header.append(f"element {name} {arr.shape[0]}")
        fields = []
        for f in arr.dtype.names:
            code = arr.dtype[f].str.lstrip("<>=|")
            if code not in _NP_TO_PLY:
                raise TypeError(f"element {name!r} property {f!r}: unsupported dtype {arr.dtype[f]}")
            header.append(f"property {_NP_TO_PLY[code]} {f}")
            fields.append((f, "<" + code))
        packed = np.empty(arr.shape[0], dtype=np.dtype(fields))   # packed, little-endian, no padding
        for f in arr.dtype.names:
            packed[f] = arr[f]
        blobs.append(packed.tobytes())
    header.append("end_header")
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        for b in blobs:
            fh.write(b)


def read_ply(path):
    """Returns a list of (name, structured ndarray) in file order."""
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    nl = data.find(b"\n", end)
    lines = data[:end].decode("ascii", "replace").splitlines()
    body = data[nl + 1:]
    fmt, elements = None, []
    for line in lines[1:]:
        tok = line.split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append([tok[1], int(tok[2]), []])
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError(f"{path}: list properties are not supported (element {elements[-1][0]})")
            if tok[1] not in _PLY_TO_NP:
                raise ValueError(f"{path}: unknown property type {tok[1]}")
            elements[-1][2].append((tok[2], _PLY_TO_NP[tok[1]]))
    if fmt not in ("binary_little_endian", "binary_big_endian", "ascii"):
        raise ValueError(f"{path}: unsupported format {fmt}")
    out = []
    if fmt == "ascii":
        tokens = body.split()
        pos = 0
        for name, count, props in elements:
            arr = np.empty(count, dtype=np.dtype([(p, "<" + t) for p, t in props]))
            n = len(props)
            block = tokens[pos:pos + count * n]
            pos += count * n
            for j, (p, t) in enumerate(props):
                col = block[j::n]
                arr[p] = np.array(col, dtype=np.float64).astype(t) if t[0] == "f" else np.array(col, dtype=np.int64).astype(t)
            out.append((name, arr))
        return out
    e = "<" if fmt == "binary_little_endian" else ">"
    off = 0
    for name, count, props in elements:
        dt = np.dtype([(p, e + t) for p, t in props])
        nbytes = dt.itemsize * count
        if off + nbytes > len(body):
            raise ValueError(f"{path}: truncated element {name}")
        arr = np.frombuffer(body, dtype=dt, count=count, offset=off)
        off += nbytes
        out.append((name, arr.astype(dt.newbyteorder("<")) if e == ">" else arr.copy()))
    return out


def element(elements, name):
    for n, a in elements:
        if n == name:
            return a
    raise KeyError(name)


def table(columns, names, dtype="f4"):
    """[N,len(names)] array -> structured array with one scalar property per column."""
    columns = np.asarray(columns)
    arr = np.empty(columns.shape[0], dtype=np.dtype([(n, dtype) for n in names]))
    for j, n in enumerate(names):
        arr[n] = columns[:, j]
    return arr
