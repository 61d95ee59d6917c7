"""TEST INFRASTRUCTURE ONLY -- live import of the upstream RAFT reference.

This module imports the *unmodified* reference (``/root/reference``) in THIS
container, with stub modules standing in for the un-vendored third-party
packages the reference imports at module level (``moorpy``, ``ccblade``) and
that are not installed here (SURVEY.md section 8c recipe).  It is used for two
things only:

  * generating the golden vectors under ``tests/golden/`` (``make_golden.py``),
  * pinning ``oracle/`` (the C / numpy restatement) against the live reference.

Nothing under ``raft_amd/`` may import this file.  On the GPU box ``/root/reference``
does not exist; there the only thing to import is the byte-compiled archive
``oracle/_ref/raft_reference.zip`` that ``oracle/stage_reference.py`` builds in the
build container (git-ignored, travels with the snapshot) -- used by ``bench.py``'s
``cpu_baseline`` leg (kind "reference-numpy") and by tests, never by the product.
"""
import os
import sys
import types
import copy

import numpy as np

REFERENCE_ROOT = os.environ.get("RAFT_REFERENCE_ROOT", "/root/reference")
ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "raft_reference.zip")


def tree_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "raft"))


def archive_available():
    return os.path.isfile(ARCHIVE)


def reference_available():
    return tree_available() or archive_available()


def reference_kind():
    """'tree' (sources under REFERENCE_ROOT: the build container), 'archive' (oracle/_ref/raft_reference.zip: the GPU
    box) or None.  RAFTX_REF_FORCE_ARCHIVE=1 picks the archive even where the tree exists (to rehearse the GPU box)."""
    if archive_available() and (os.environ.get("RAFTX_REF_FORCE_ARCHIVE") or not tree_available()):
        return "archive"
    return "tree" if tree_available() else None


def _raise_stub(*a, **k):
    raise RuntimeError("third-party stub called (moorpy/ccblade are not installed)")


def install_stubs():
    """Register stub modules for moorpy / ccblade (imported at module level by
    raft/raft_model.py:17,20  raft/raft_fowt.py:13-14  raft/raft_member.py:8
    raft/raft_rotor.py:18-21) and put the reference on sys.path."""
    if "moorpy" not in sys.modules:
        mp = types.ModuleType("moorpy")
        mph = types.ModuleType("moorpy.helpers")
        for n in ("dsolve2", "set_axes_equal", "dsolvePlot", "lines2ss",
                  "transformPosition"):
            setattr(mph, n, _raise_stub)
        mp.helpers = mph
        mp.System = type("System", (), {"__init__": _raise_stub})
        sys.modules["moorpy"] = mp
        sys.modules["moorpy.helpers"] = mph
    if "ccblade" not in sys.modules:
        cc = types.ModuleType("ccblade")
        ccc = types.ModuleType("ccblade.ccblade")
        ccc.CCBlade = ccc.CCAirfoil = type("X", (), {"__init__": lambda s, *a, **k: None})
        cc.ccblade = ccc
        sys.modules["ccblade"] = cc
        sys.modules["ccblade.ccblade"] = ccc
    if "pyhams" not in sys.modules:
        # pyHAMS (un-vendored, absent) is only asked for its two WAMIT-file readers by FOWT.readHydro
        # (raft_fowt.py:1452-1456): raft_amd/bem.py provides them, so the REFERENCE's readHydro / BEM excitation run
        # unmodified on top of our parsers (the parsers themselves stay "parity unpinned", see raft_amd/bem.py)
        from raft_amd import bem
        ph = types.ModuleType("pyhams")
        php = types.ModuleType("pyhams.pyhams")
        php.read_wamit1, php.read_wamit3 = bem.read_wamit1, bem.read_wamit3
        ph.pyhams = php
        sys.modules["pyhams"] = ph
        sys.modules["pyhams.pyhams"] = php
    root = ARCHIVE if reference_kind() == "archive" else REFERENCE_ROOT
    if root not in sys.path:
        sys.path.insert(0, root)


def import_raft():
    if not reference_available():
        raise RuntimeError("reference not present: neither %s nor %s" % (REFERENCE_ROOT, ARCHIVE))
    install_stubs()
    import matplotlib
    matplotlib.use("Agg")
    import raft  # noqa: F401
    return raft


def load_design(path):
    import yaml
    with open(path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def load_deck(rel):
    """A deck of the reference tree by its path relative to the tree ('examples/VolturnUS-S_example.yaml'); read from the
    staged archive (decks/<basename>) when that is what this host has."""
    if reference_kind() == "archive":
        import yaml
        import zipfile
        with zipfile.ZipFile(ARCHIVE) as z:
            return yaml.load(z.read("decks/" + os.path.basename(rel)), Loader=yaml.FullLoader)
    return load_design(os.path.join(REFERENCE_ROOT, rel))


DEFAULT_C_MOOR = np.diag([7e4, 7e4, 0.0, 0.0, 0.0, 1e8])


def prepare_design(design, settings=None):
    """Strip the parts that need MoorPy / CCBlade: no mooring
    system, aero-servo off."""
    design = copy.deepcopy(design)
    design.pop("mooring", None)
    design.pop("array_mooring", None)
    for key in ("turbine",):
        if key in design and design[key] is not None:
            design[key]["aeroServoMod"] = 0
    if "turbines" in design:
        for t in design["turbines"]:
            t["aeroServoMod"] = 0
    if settings:
        design.setdefault("settings", {})
        design["settings"].update(settings)
    return design


def build_model(design, case=None, c_moor=None, r6=None):
    """Model(design) -> setPosition -> calcStatics -> calcHydroConstants ->
    calcTurbineConstants, with an injected C_moor.  Mirrors what
    analyzeCases/solveStatics leave behind before solveDynamics is called
    (raft_model.py:277-283, :602, :620)."""
    raft = import_raft()
    model = raft.Model(design)
    if case is None:
        case = dict(wind_speed=0, wind_heading=0, turbulence=0,
                    turbine_status="off", yaw_misalign=0,
                    wave_spectrum="JONSWAP", wave_period=12, wave_height=6,
                    wave_heading=0)
    for i, fowt in enumerate(model.fowtList):
        pose = np.zeros(fowt.nDOF) if r6 is None else np.array(r6[i], dtype=float)
        if r6 is None:
            pose[0] = fowt.x_ref
            pose[1] = fowt.y_ref
        fowt.setPosition(pose)
        fowt.calcStatics()
        fowt.calcHydroConstants()
        fowt.calcTurbineConstants(dict(case), ptfm_pitch=0)
        fowt.C_moor = np.array(DEFAULT_C_MOOR if c_moor is None else c_moor, dtype=float)
    return model


def make_case(Hs=6.0, Tp=12.0, heading=0.0, spectrum="JONSWAP", gamma=0):
    return dict(wind_speed=0, wind_heading=0, turbulence=0, turbine_status="off",
                yaw_misalign=0, wave_spectrum=spectrum, wave_period=Tp,
                wave_height=Hs, wave_heading=heading, wave_gamma=gamma,
                current_speed=0, current_heading=0)
