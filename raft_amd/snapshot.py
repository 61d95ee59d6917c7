"""Snapshots of the reference's Model / FOWT / Member objects: the attributes the hot path reads, as plain arrays.

``snapshot_model`` records, from a live reference Model, exactly the attributes raft_amd reads (raft_amd/strips.py,
raft_amd/dropin.py); ``build_model`` rebuilds duck-typed attribute containers from such a record; ``save_fixture`` /
``load_fixture`` (de)serialise nested dict / list / array trees into one ``.npz`` file (no pickle).  The parity tests,
bench.py and the examples use it to run where the reference tree is absent (the GPU box has no /root/reference); a RAFT
user can use it to ship a model's hot-path inputs to a GPU node.
"""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

MEMBER_ARRAYS = ("r", "q", "p1", "p2", "ds", "drs", "dls", "ls", "stations",
                 "Cd_q", "Cd_p1", "Cd_p2", "Cd_End", "Ca_q", "Ca_p1", "Ca_p2", "Ca_End")
FOWT_ARRAYS = ("T", "w", "k", "M_struc", "B_struc", "C_struc", "C_hydro", "C_moor", "C_elast",
               "A_hydro_morison", "B_gyro")
FOWT_SCALARS = ("nDOF", "nFullDOF", "rho_water", "g", "depth", "dw", "nw", "x_ref", "y_ref",
                "heading_adjust", "potModMaster", "potSecOrder", "nrotors")


class Obj:
    def __repr__(self):
        return "Obj(%s)" % ", ".join(sorted(self.__dict__))


# ------------------------------------------------------------------ snapshot
def snapshot_member(mem):
    d = {"name": str(mem.name), "type": str(mem.type), "shape": str(mem.shape),
         "potMod": bool(mem.potMod), "MCF": bool(mem.MCF), "ns": int(mem.ns)}
    for a in MEMBER_ARRAYS:
        d[a] = np.array(getattr(mem, a), dtype=float)
    d["rA"] = np.array(mem.rA, dtype=float)
    d["rB"] = np.array(mem.rB, dtype=float)
    d["node_r"] = np.array(mem.nodeList[0].r, dtype=float)
    d["node_T"] = np.array(mem.nodeList[0].T, dtype=float)
    # every structural node of the member (flexible members have one per strip): position and number in the unit's
    # full-DOF vector -- what the node-by-node path of units with more than 6 reduced DOFs reads
    d["nodes_r"] = np.array([np.asarray(n.r, dtype=float)[:3] for n in mem.nodeList])
    d["nodes_id"] = np.array([int(n.id) for n in mem.nodeList], dtype=np.int64)
    # reference products kept for packer checks (not read by raft_amd)
    d["ref_Imat"] = np.array(mem.Imat, dtype=float)
    d["ref_a_i"] = np.array(mem.a_i, dtype=float)
    if mem.MCF:
        d["ref_Imat_MCF"] = np.array(mem.Imat_MCF, dtype=complex)
    return d


def snapshot_fowt(fowt):
    d = {}
    for a in FOWT_ARRAYS:
        d[a] = np.array(getattr(fowt, a), dtype=float)
    for a in FOWT_SCALARS:
        d[a] = getattr(fowt, a)
    d["potMod"] = bool(fowt.potMod)
    for a in ("A_BEM", "B_BEM", "A_aero", "B_aero"):
        v = np.asarray(getattr(fowt, a))
        if np.any(v):
            d[a] = np.array(v, dtype=float)
        else:
            d[a + "_zero_shape"] = np.array(v.shape, dtype=np.int64)
    if hasattr(fowt, "X_BEM"):
        d["X_BEM"] = np.array(fowt.X_BEM, dtype=complex)
        d["BEM_headings"] = np.array(fowt.BEM_headings, dtype=float)
    if getattr(fowt, "potSecOrder", 0) == 1:
        d["w1_2nd"] = np.array(fowt.w1_2nd, dtype=float)
        d["k1_2nd"] = np.array(fowt.k1_2nd, dtype=float)
    d["members"] = [snapshot_member(m) for m in fowt.memberList]
    return d


def snapshot_model(model):
    return {"nw": int(model.nw), "nIter": int(model.nIter), "XiStart": float(model.XiStart),
            "nDOF": int(model.nDOF), "w": np.array(model.w), "depth": float(model.depth),
            "fowts": [snapshot_fowt(f) for f in model.fowtList]}


# ------------------------------------------------------------------ (de)serialise
def _flatten(obj, prefix, arrays, meta):
    if isinstance(obj, dict):
        m = {}
        for k, v in obj.items():
            m[k] = _flatten(v, prefix + "/" + str(k), arrays, meta)
        return {"__dict__": m}
    if isinstance(obj, (list, tuple)):
        return {"__list__": [_flatten(v, prefix + "/" + str(i), arrays, meta) for i, v in enumerate(obj)]}
    if isinstance(obj, np.ndarray):
        arrays[prefix] = obj
        return {"__array__": prefix}
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (np.bool_,)):
        return bool(obj)
    if isinstance(obj, (bool, int, float, str)) or obj is None:
        return obj
    raise TypeError("cannot serialise %r at %s" % (type(obj), prefix))


def save_fixture(path, obj):
    arrays, meta = {}, {}
    tree = _flatten(obj, "", arrays, meta)
    arrays["__tree__"] = np.array(json.dumps(tree))
    np.savez_compressed(path, **arrays)


def _unflatten(node, z):
    if isinstance(node, dict):
        if "__dict__" in node:
            return {k: _unflatten(v, z) for k, v in node["__dict__"].items()}
        if "__list__" in node:
            return [_unflatten(v, z) for v in node["__list__"]]
        if "__array__" in node:
            return z[node["__array__"]]
    return node


def load_fixture(name):
    path = name if os.path.isabs(name) else os.path.join(GOLDEN_DIR, name)
    with np.load(path, allow_pickle=False) as z:
        tree = json.loads(str(z["__tree__"]))
        return _unflatten(tree, z)


# ------------------------------------------------------------------ rebuild
def build_member(d):
    m = Obj()
    for k in ("name", "type", "shape", "potMod", "MCF", "ns"):
        setattr(m, k, d[k])
    for a in MEMBER_ARRAYS:
        setattr(m, a, d[a])
    if "rA" in d:
        m.rA, m.rB = d["rA"], d["rB"]
    node = Obj()
    node.r = d["node_r"]
    node.T = d["node_T"]
    node.id = 0
    node.nDOF = 6
    m.nodeList = [node]
    if "nodes_id" in d:                                   # fixtures written since the node-by-node path exists
        node.id = int(d["nodes_id"][0])
        for r, i in zip(d["nodes_r"][1:], d["nodes_id"][1:]):
            nd = Obj()
            nd.r, nd.id, nd.nDOF = r, int(i), 6
            m.nodeList.append(nd)
    m.ref = {k[4:]: v for k, v in d.items() if k.startswith("ref_")}
    return m


def build_fowt(d):
    f = Obj()
    for a in FOWT_ARRAYS:
        setattr(f, a, d[a])
    for a in FOWT_SCALARS:
        setattr(f, a, d[a])
    f.potMod = d["potMod"]
    for a in ("A_BEM", "B_BEM", "A_aero", "B_aero"):
        if a in d:
            setattr(f, a, d[a])
        else:
            setattr(f, a, np.zeros(tuple(int(x) for x in d[a + "_zero_shape"])))
    if "X_BEM" in d:
        f.X_BEM, f.BEM_headings = d["X_BEM"], d["BEM_headings"]
    if "w1_2nd" in d:
        f.w1_2nd, f.k1_2nd = d["w1_2nd"], d["k1_2nd"]
    f.memberList = [build_member(m) for m in d["members"]]
    f.rotorList = []
    f.ms = None
    f.moorMod = 0
    return f


def build_model(d):
    m = Obj()
    m.nw, m.nIter, m.XiStart, m.nDOF = d["nw"], d["nIter"], d["XiStart"], d["nDOF"]
    m.w = d["w"]
    m.depth = d["depth"]
    m.fowtList = [build_fowt(f) for f in d["fowts"]]
    m.ms = None
    m.results = {}
    return m


def case_from_fixture(c):
    """The load-case dict of a fixture case, as a fresh (deep) copy with lists for the per-heading entries."""
    import copy
    return copy.deepcopy({k: (list(v) if isinstance(v, (list, np.ndarray)) else v) for k, v in c["case"].items()})


def ref_headings(c):
    """(reference responses of the wave headings [nH,6N,nw], nH) of a fixture case.  Full cases store the reference's
    Xi with its zero rotor-excitation row (raft_model.py:1236), lean ones (many-case fixtures) without it."""
    nH = len(np.atleast_1d(np.asarray(c["case"]["wave_heading"], dtype=float)))
    return np.asarray(c["Xi"])[:nH], nH


def load_model_fixture(name):
    fx = load_fixture(name)
    return fx, build_model(fx["model"])
