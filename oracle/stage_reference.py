"""TEST / BASELINE INFRASTRUCTURE ONLY -- "builds" the pure-Python reference for the cpu_baseline leg of bench.py.

The reference (WISDEM/RAFT, /root/reference) has no compiled part, and /root/reference does not exist on the GPU box.
What a C reference gets from `gcc -shared` into oracle/_ref/ a Python reference gets from `py_compile`: this recipe
byte-compiles the modules of the reference's `raft` package that `raft.Model.solveDynamics` (raft/raft_model.py:966)
imports, FROM THE SOURCES WHERE THEY LIE under /root/reference, into ONE archive

    oracle/_ref/raft_reference.zip        raft/<module>.pyc  (no .py sources), decks/<deck>.yaml, MANIFEST.json

`oracle/_ref/` is git-ignored (it never enters history) and not gpurun-ignored (it travels to the GPU box with the
snapshot, like the built .so files).  The archive is importable as it is (`zipimport` reads sourceless .pyc at the
legacy locations), so `oracle/ref_harness.py` can put it on sys.path when /root/reference is absent.

Who may use it: `bench.py`'s `cpu_baseline` leg (kind "reference-numpy") and tests -- never `raft_amd/`.

usage: python oracle/stage_reference.py            (also called by __graft_entry__.build() when /root/reference exists)
"""
import hashlib
import json
import os
import py_compile
import sys
import tempfile
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("RAFT_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "raft_reference.zip")

# the import closure of raft/__init__.py -> raft_model (raft_model.py:17-21, raft_fowt.py:8-14, raft_rotor.py:8-16)
MODULES = ("__init__", "raft_model", "raft_fowt", "raft_member", "raft_node", "raft_rotor", "helpers", "member2pnl",
           "pyIECWind")
# input decks the timed workloads read (BASELINE.json configs 1-3; data, not code)
DECKS = ("examples/VolturnUS-S_example.yaml", "designs/OC3spar.yaml")


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def stage(verbose=True):
    src_pkg = os.path.join(REFERENCE_ROOT, "raft")
    if not os.path.isdir(src_pkg):
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest = {"python": "%d.%d.%d" % sys.version_info[:3], "reference_root": REFERENCE_ROOT, "modules": {}, "decks": {}}
    tmp = tempfile.mkdtemp(prefix="raftref_")
    part = ARCHIVE + ".part"
    with zipfile.ZipFile(part, "w", zipfile.ZIP_DEFLATED) as z:
        for m in MODULES:
            src = os.path.join(src_pkg, m + ".py")
            pyc = os.path.join(tmp, m + ".pyc")
            # dfile: the path tracebacks show; UNCHECKED_HASH: no mtime / source lookup at import time
            py_compile.compile(src, cfile=pyc, dfile="<reference>/raft/%s.py" % m, doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
            z.write(pyc, "raft/%s.pyc" % m)
            manifest["modules"][m] = _sha(src)
        for d in DECKS:
            src = os.path.join(REFERENCE_ROOT, d)
            z.write(src, "decks/" + os.path.basename(d))
            manifest["decks"][os.path.basename(d)] = _sha(src)
        z.writestr("MANIFEST.json", json.dumps(manifest, indent=1))
    os.replace(part, ARCHIVE)
    if verbose:
        print("staged %s (%d modules, %d decks, %d bytes)" % (ARCHIVE, len(MODULES), len(DECKS), os.path.getsize(ARCHIVE)))
    return ARCHIVE


if __name__ == "__main__":
    stage()
