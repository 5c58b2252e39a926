"""What the compiler allocated for the kernels of the built library, read from the code object's metadata (no GPU needed): the
isotropic kernels and the scalar-ODE dense kernels -- the lane layout of csrc/pnode_lanes.hpp -- must not use scratch memory for any
order the library instantiates (nu <= 8). Round 3's per-lane-redundant kernels spilled from nu = 4 on and used up to 3 KB of scratch per
lane at nu = 8; one of those spills was the memory access fault DESIGN.md section 6b explains."""
import os
import re
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd", "csrc", "libpnode_hip.so")


def _elf_notes(blob):
    """msgpack metadata blobs (NT_AMDGPU_METADATA = 32) of a 64-bit little-endian ELF image."""
    shoff = struct.unpack_from("<Q", blob, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", blob, 0x3A)
    out = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        if sh + 64 > len(blob):
            break
        sh_type, = struct.unpack_from("<I", blob, sh + 4)
        off, size = struct.unpack_from("<QQ", blob, sh + 0x18)
        if sh_type != 7:  # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", blob, p)
            p += 12
            name = blob[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = blob[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                out.append(desc)
    return out


def kernel_metadata(path=LIB):
    """{kernel symbol: metadata dict} over every gfx950 code object embedded in the shared library."""
    msgpack = pytest.importorskip("msgpack")
    data = open(path, "rb").read()
    kernels = {}
    pos = 0
    while True:
        pos = data.find(b"\x7fELF\x02\x01\x01", pos)
        if pos < 0:
            break
        machine, = struct.unpack_from("<H", data, pos + 0x12)
        if machine == 224:  # EM_AMDGPU
            for desc in _elf_notes(data[pos:]):
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    kernels[k[".name"]] = k
        pos += 4
    return kernels


def test_lane_layout_kernels_use_no_scratch():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    ks = kernel_metadata()
    iso = {n: k for n, k in ks.items() if "iso_" in n or "dense_d1_" in n}
    assert len(iso) >= 150, len(iso)  # step / advance / interp / smoother / off-grid / sample kernels for n = 2 .. 9
    bad = {n: k[".private_segment_fixed_size"] for n, k in iso.items() if k[".private_segment_fixed_size"] != 0}
    assert not bad, bad
    # the filter's step kernels (host controller) stay far from the register limit at every order
    for n, k in iso.items():
        if re.search(r"iso_step_kernelILi\d+ELb0E", n):  # FP = false
            assert k[".vgpr_count"] <= 160 and k.get(".agpr_count", 0) == 0, (n, k[".vgpr_count"], k.get(".agpr_count"))
