"""Gauge-configuration file formats of the reference (host logic, numpy only).

The reference loads/saves configurations through Gaugefields.jl (`load_BridgeText!`,
`ILDG_format`; call sites /root/reference/src/system/universe.jl:58-77, src/system/lqcd.jl:226-247).
On-disk order verified on the reference's fixtures, slowest -> fastest:
    t, z, y, x, mu(=x,y,z,t), a (row), b (column), (re, im)
In memory the reference holds U[mu][a,b,x,y,z,t] (a fastest), so the 3x3 block is transposed on load.
Returned arrays are numpy complex128, C order, shape (4, NT, NZ, NY, NX, NC, NC) indexed
U[mu,t,z,y,x,b,a]  -- the same memory image as the Julia arrays.
"""
import struct

import numpy as np

LIME_MAGIC = 0x456789AB


def _disk_to_memory(flat, L, NC):
    NX, NY, NZ, NT = L
    d = flat.reshape(NT, NZ, NY, NX, 4, NC, NC)           # [t,z,y,x,mu,a,b]
    return np.ascontiguousarray(d.transpose(4, 0, 1, 2, 3, 6, 5))  # [mu,t,z,y,x,b,a]


def _memory_to_disk(U):
    return np.ascontiguousarray(U.transpose(1, 2, 3, 4, 0, 6, 5)).reshape(-1)  # [t,z,y,x,mu,a,b]


def load_BridgeText(path, L, NC=3):
    """One real number per line, 2*NC^2*4*V lines (Bridge++ text format)."""
    V = L[0] * L[1] * L[2] * L[3]
    vals = np.loadtxt(path, dtype=np.float64)
    if vals.size != 2 * NC * NC * 4 * V:
        raise ValueError(f"BridgeText: expected {2 * NC * NC * 4 * V} numbers, file has {vals.size}")
    return _disk_to_memory(vals.view(np.complex128), L, NC)


def save_BridgeText(path, U):
    flat = _memory_to_disk(U).view(np.float64)
    np.savetxt(path, flat, fmt="%.15e")


def read_lime_records(buf):
    """Yield (type_string, payload_bytes) for every LIME record in buf."""
    off = 0
    while off + 144 <= len(buf):
        magic, version, flags, length = struct.unpack(">IHHQ", buf[off:off + 16])
        if magic != LIME_MAGIC:
            raise ValueError("not a LIME record (bad magic)")
        rtype = buf[off + 16:off + 144].split(b"\0", 1)[0].decode("ascii")
        payload = buf[off + 144:off + 144 + length]
        yield rtype, payload
        off += 144 + ((length + 7) // 8) * 8


def load_ildg(path, L, NC=3):
    """ILDG/LIME: big-endian fp64 payload of the `ildg-binary-data` record, same flat order as BridgeText."""
    V = L[0] * L[1] * L[2] * L[3]
    with open(path, "rb") as f:
        buf = f.read()
    for rtype, payload in read_lime_records(buf):
        if rtype == "ildg-binary-data":
            if len(payload) != V * 4 * NC * NC * 16:
                raise ValueError(f"ildg-binary-data: expected {V * 4 * NC * NC * 16} bytes, got {len(payload)}")
            vals = np.frombuffer(payload, dtype=">f8").astype(np.float64)
            return _disk_to_memory(vals.view(np.complex128), L, NC)
    raise ValueError("no ildg-binary-data record found")


def save_ildg(path, U):
    """Single-record LIME file as the reference writes it (144-byte header, flags 0xc000)."""
    payload = _memory_to_disk(U).view(np.float64).astype(">f8").tobytes()
    hdr = struct.pack(">IHHQ", LIME_MAGIC, 1, 0xC000, len(payload)) + b"ildg-binary-data".ljust(128, b"\0")
    with open(path, "wb") as f:
        f.write(hdr + payload + b"\0" * ((-len(payload)) % 8))
