"""What the gfx950 code object of the built library says about its kernels (no GPU needed: hipcc cross-compiles here and the
same .so travels to the GPU box).

* No scratch on the specialisations a default call path reaches: `.private_segment_fixed_size` of the
  code-object notes.
* The persistent fused kernels (raftx_kernels.h raftx_kp_f*) re-enter themselves at their first instruction with the
  registers a fresh dispatch would find; the register assignment they restore is the one their kernel descriptors must
  declare -- user SGPRs s[0:1] queue pointer, s[2:3] kernarg segment, system SGPRs workgroup id x, y, z, packed
  work-item ids in v0, no private segment.  A compiler that lays them out differently fails HERE, not on the GPU.
"""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "raft_amd", "csrc", "libraftx_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"

pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "clang-offload-bundler"))),
                                reason="needs the built library and the ROCm LLVM tools")


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    d = tmp_path_factory.mktemp("co")
    fat, co = str(d / "fat.bin"), str(d / "gfx950.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section=.hip_fatbin=" + fat, LIB, str(d / "unused.o")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return co


def kernel_notes(co):
    """name -> dict of the scalar fields of the code-object notes (amdhsa.kernels)."""
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S.*)$", line)
        if line.lstrip().startswith("- .agpr_count") or line.lstrip().startswith("- .args"):
            cur = {}
        if m and cur is not None:
            cur[m.group(1)] = m.group(2).strip()
            if m.group(1) == "name" and not line.lstrip().startswith("- "):
                pass
        if m and m.group(1) == "wavefront_size" and cur is not None:
            # (the last scalar field of a kernel entry in LLVM's emission order)
            if "symbol" in cur:
                out[cur["symbol"][:-3]] = cur
            cur = None
    return out


def kernel_descriptors(co):
    """name -> (compute_pgm_rsrc2, kernel_code_properties) of every <name>.kd symbol."""
    with open(co, "rb") as f:
        blob = f.read()
    assert blob[:4] == b"\x7fELF" and blob[4] == 2
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize) for i in range(shnum)]
    symtab = next(s for s in secs if s[1] == 2)                           # SHT_SYMTAB
    strtab = secs[symtab[6]]
    out = {}
    for i in range(symtab[5] // 24):
        st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", blob, symtab[4] + i * 24)
        end = blob.index(b"\0", strtab[4] + st_name)
        name = blob[strtab[4] + st_name:end].decode()
        if not name.endswith(".kd") or st_shndx == 0 or st_shndx >= len(secs):
            continue
        sec = secs[st_shndx]
        off = sec[4] + (st_value - sec[3])
        rsrc2, = struct.unpack_from("<I", blob, off + 52)
        props, = struct.unpack_from("<H", blob, off + 56)
        out[name[:-3]] = (rsrc2, props)
    return out


def test_persistent_kernels_entry_state(code_object):
    notes, kds = kernel_notes(code_object), kernel_descriptors(code_object)
    names = sorted(n for n in kds if n.startswith("raftx_kp_f"))
    assert len(names) == 12, names                                       # RAFTX_PERSIST128: one twin per lean specialisation
    assert "raftx_kpg_f0" in kds                                         # the generating form (raftx_fusedgen.h) re-enters the same way
    for n in names + ["raftx_kpg_f0"]:
        rsrc2, props = kds[n]
        assert (props & 0x7F) == 0b0001100, (n, bin(props))              # QUEUE_PTR + KERNARG_SEGMENT_PTR, nothing else
        assert ((rsrc2 >> 1) & 0x1F) == 4, (n, "user SGPR count")        # s[0:1] queue, s[2:3] kernarg
        assert ((rsrc2 >> 7) & 0xF) == 0b0111, (n, "workgroup id x, y, z in s4..s6, no workgroup info")
        assert ((rsrc2 >> 11) & 3) == 2, (n, "work-item ids x, y, z packed in v0")
        # (gfx950 has architected flat scratch: a private segment adds no SGPR to the entry state, so the featured twins
        # that spill a few dwords re-enter correctly too; the plain ones must not spill at all -- NO_SCRATCH below)
        assert int(notes[n]["private_segment_fixed_size"]) <= 128, n
        assert notes[n]["uses_dynamic_stack"] == "false", n
        assert int(notes[n]["vgpr_count"]) <= 256 and int(notes[n]["agpr_count"]) == 0, n


# kernels behind the default shapes of section-8 rows that must not spill to scratch
NO_SCRATCH = [
    r"^raftx_kp_f(0|4|16)$",                                             # persistent fused fixed point: plain, F_wave out, MacCamy-Fuchs
    r"^_Z16k_solve_dynamicsILi2ELi0ELi128ELi2EE",                        # the one-workgroup-per-pair lean kernel (C3)
    r"^_Z12k_excitationILi2ELi128ELi2EE",                                # calcHydroExcitation at the 200-bin shape
    r"^_Z11k_linearizeILi2ELi128ELi2EE",                                 # calcHydroLinearization at the 200-bin shape
    r"^_Z16k_solve_dynamicsILi2ELi(1|4|9|16|17|32|36|48|49)ELi128ELi2EE",   # lean featured sweeps (per-pair launches)
    r"^raftx_kp_f(1|9|17|48|49)$",                                       # ... and their persistent twins
    r"^_Z11k_qtf_pairs",                                                 # C5
]


def test_no_scratch_on_default_shapes(code_object):
    notes = kernel_notes(code_object)
    assert len(notes) > 100
    hit = {pat: 0 for pat in NO_SCRATCH}
    bad = []
    for name, k in notes.items():
        for pat in NO_SCRATCH:
            if re.search(pat, name):
                hit[pat] += 1
                if int(k["private_segment_fixed_size"]) != 0:
                    bad.append((name, int(k["private_segment_fixed_size"])))
    assert all(hit.values()), hit
    assert not bad, bad
