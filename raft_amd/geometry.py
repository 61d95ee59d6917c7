"""Member descriptors: the host-side feeder of ``raftx_build_designs``.

The reference turns every entry of a design's ``platform: members`` list (plus
towers / nacelles) into ``Member`` objects, discretises them into strips and
evaluates per-strip constants in Python (0.16 s per design).
Here the host only PARSES the member description -- a few dozen numbers per
member -- and the device does the rest (``raftx_build_designs``,
include/raftx.h): strip discretisation, pose, hydrodynamic constants, Morison
added mass, hydrostatics.

This file restates the input handling of
    raft/raft_member.py:36-190   Member.__init__ (end points, heading, stations,
                                  diameters / side pairs, coefficients, ballast)
    raft/raft_fowt.py:223-272    FOWT.__init__ (potModMaster, dlsMax default,
                                  one Member per heading; towers, nacelles)
    raft/helpers.py:828-915      getFromDict (scalar / list broadcasting rules)
and nothing else: no strips are built on the host.

Record layouts (include/raftx.h): member RAFTX_GM_* (16 doubles), station
RAFTX_GS_* (16 doubles).
"""
import numpy as np

GM_N, GS_N, GC_N = 16, 16, 4
GM_RA, GM_RB, GM_GAMMA, GM_SHAPE, GM_DLSMAX, GM_FLAGS, GM_L, GM_RHOSHELL = 0, 3, 6, 7, 8, 9, 10, 11
GS_S, GS_D, GS_T, GS_CD, GS_CA, GS_LFILL, GS_RHOFILL = 0, 1, 3, 4, 8, 12, 13
FLAG_POTMOD, FLAG_MCF, FLAG_NOSTATIC = 1, 2, 4
ADD_MORISON, ADD_HYDROSTATIC, ADD_INERTIA, TRIM_BALLAST = 1, 2, 4, 8
SP_N, SP_V, SP_AWP, SP_RCB, SP_MASS, SP_RCG, SP_DRHO, SP_VFILL = 12, 0, 1, 2, 5, 6, 9, 10


class UnsupportedMember(Exception):
    """Member outside the device generator's scope (flexible 'beam' members)."""


# ---- reading a member entry of the YAML.  The deck format is upstream's (a member lists its stations and gives
# every per-station quantity either once or once per station; coefficient pairs are [p1, p2], helpers.py:828-915 is the
# reader upstream applies); each kind of entry has its own small reader here.
class DeckError(ValueError):
    pass


def _scalar(entry, key, default=None, kind=float):
    """A single value; a list where one value is expected is an error of the deck."""
    if key not in entry:
        if default is None:
            raise DeckError("Key '%s' not found in input file..." % key)
        return default
    v = entry[key]
    if not np.isscalar(v):
        raise DeckError("Value for key '%s' is expected to be a scalar but instead is: %s" % (key, v))
    return kind(v)


def _as_given(entry, key, default):
    """A value of whatever shape the deck gives it (scalar or list)."""
    if key not in entry:
        return default
    v = entry[key]
    return float(v) if np.isscalar(v) else np.array(v, dtype=float)


def _per_station(entry, key, n, default=None):
    """One value per station: a scalar applies to all n stations, a list must have n items."""
    if key not in entry:
        if default is None:
            raise DeckError("Key '%s' not found in input file..." % key)
        return np.tile(default, n)
    v = entry[key]
    if np.isscalar(v):
        return np.tile(float(v), n)
    if len(v) != n:
        raise DeckError("Value for key '%s' is not the expected size of %s and is instead: %s" % (key, n, v))
    return np.array([float(x) for x in v])


def _per_station_side(entry, key, n, side, default):
    """Coefficient of cross-section axis ``side`` (0: p1, 1: p2) per station.  The deck may give one number, one
    [p1, p2] pair for the whole member (a flat list of length n = 2 is read that way, as upstream reads it), or one
    pair per station."""
    if key not in entry:
        return np.tile(default, n)
    v = entry[key]
    if np.isscalar(v):
        return np.tile(float(v), n)
    if len(v) != n:
        raise DeckError("Value for key '%s' is not the expected size of %s and is instead: %s" % (key, n, v))
    a = np.array(v)
    width = a.shape[0] if a.ndim == 1 else a.shape[1]
    if side not in range(width):
        raise DeckError("Value for index '%s' is not within the size of %s" % (side, v))
    return np.tile(v[side], n) if a.ndim == 1 else np.array([row[side] for row in v])


def _per_station_pair(entry, key, n):
    """[n, 2] side lengths of a rectangular member: one [a, b] pair for all stations or one per station."""
    if key not in entry:
        raise DeckError("Key '%s' not found in input file..." % key)
    v = entry[key]
    if np.isscalar(v):
        return np.tile(float(v), [n, 2])
    a = np.array(v, dtype=float)
    if list(a.shape) == [n, 2]:
        return a
    if a.ndim == 1 and len(a) == 2:
        return np.tile(a, [n, 1])
    raise DeckError("Value for key '%s' is not a compatible size for target size of %s" % (key, [n, 2]))


def _heading(r, heading):
    """helpers.py:587-602 applyHeadingToPoint"""
    if heading == 0.0:
        return r
    c, s = np.cos(np.deg2rad(heading)), np.sin(np.deg2rad(heading))
    return np.matmul(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]), r)


def describe_member(mi, heading=0.0, part_of="platform"):
    """(gm [16], gs [n,16], caps [ncap,4]) of one member copy -- raft_member.py:36-190."""
    mtype = str(mi.get("type", "rigid"))
    if mtype != "rigid":
        raise UnsupportedMember("member '%s' is type '%s'; only rigid members are generated on the device"
                                % (mi.get("name", "?"), mtype))
    rA0 = np.array(mi["rA"], dtype=np.double)
    rB0 = np.array(mi["rB"], dtype=np.double)
    if rA0[2] == 0 or rB0[2] == 0:
        raise ValueError("RAFT Members cannot start or end on the waterplane")
    shape = str(mi["shape"])
    gamma = _scalar(mi, "gamma", default=0.)
    rAB = rB0 - rA0
    length = np.linalg.norm(rAB)
    if heading != 0.0:
        rA0 = _heading(rA0, heading)
        rB0 = _heading(rB0, heading)
        if rAB[0] == 0.0 and rAB[1] == 0:
            gamma += heading
    st = np.array(mi["stations"], dtype=float)
    n = len(st)
    if n < 2:
        raise ValueError("At least two stations entries must be provided")
    if not sorted(st) == st.tolist():
        raise ValueError("Member %s: the station list is not in ascending order." % mi.get("name", "?"))
    gs = np.zeros((n, GS_N))
    gs[:, GS_S] = (st - st[0]) / (st[-1] - st[0]) * length
    if shape[0].lower() == "c":
        circ = True
        d = _per_station(mi, "d", n)
        gs[:, GS_D] = d
        gs[:, GS_D + 1] = d
        gamma = 0
    elif shape[0].lower() == "r":
        circ = False
        gs[:, GS_D:GS_D + 2] = _per_station_pair(mi, "d", n)
    else:
        raise ValueError("The only allowable shape strings are circular and rectangular")
    mcf = bool(_scalar(mi, "MCF", default=False, kind=bool)) and circ
    potmod = bool(_scalar(mi, "potMod", default=False, kind=bool))
    gs[:, GS_T] = _per_station(mi, "t", n, default=0)
    st_fill = _per_station(mi, "l_fill", n - 1, default=0)
    for i in range(n - 1):
        if st_fill[i] < 0:
            raise Exception("Member %s: ballast level in section %d is negative." % (mi.get("name", "?"), i + 1))
        if st_fill[i] > st[i + 1] - st[i]:
            raise Exception("Member %s: ballast level in section %d exceeds section length." % (mi.get("name", "?"), i + 1))
    gs[:n - 1, GS_LFILL] = st_fill / (st[-1] - st[0]) * length
    rho_fill = _as_given(mi, "rho_fill", 1025)
    if np.isscalar(rho_fill):
        gs[:n - 1, GS_RHOFILL] = rho_fill
    elif len(rho_fill) == n - 1:
        gs[:n - 1, GS_RHOFILL] = np.array(rho_fill)
    else:
        raise Exception("Member %s: the number of provided ballast densities (rho_fill) must be 1 less than the "
                        "number of stations." % mi.get("name", "?"))
    gs[:, GS_CD + 0] = _per_station(mi, "Cd_q", n, default=0.0)
    gs[:, GS_CD + 1] = _per_station_side(mi, "Cd", n, 0, 0.6)
    gs[:, GS_CD + 2] = _per_station_side(mi, "Cd", n, 1, 0.6)
    gs[:, GS_CD + 3] = _per_station(mi, "CdEnd", n, default=0.6)
    gs[:, GS_CA + 0] = _per_station(mi, "Ca_q", n, default=0.0)
    gs[:, GS_CA + 1] = _per_station_side(mi, "Ca", n, 0, 0.97)
    gs[:, GS_CA + 2] = _per_station_side(mi, "Ca", n, 1, 0.97)
    gs[:, GS_CA + 3] = _per_station(mi, "CaEnd", n, default=0.6)
    gm = np.zeros(GM_N)
    gm[GM_RA:GM_RA + 3] = rA0
    gm[GM_RB:GM_RB + 3] = rB0
    gm[GM_GAMMA] = gamma
    gm[GM_SHAPE] = 1.0 if circ else 0.0
    gm[GM_DLSMAX] = _scalar(mi, "dlsMax", default=5)
    gm[GM_FLAGS] = (FLAG_POTMOD if potmod else 0) | (FLAG_MCF if mcf else 0) | \
                   (FLAG_NOSTATIC if part_of == "nacelle" else 0)
    gm[GM_L] = length
    gm[GM_RHOSHELL] = _scalar(mi, "rho_shell", default=8500.)
    # end caps / bulkheads (raft_member.py:162-175)
    cap_st = _as_given(mi, "cap_stations", [])
    cap_st = np.atleast_1d(np.array(cap_st, dtype=float))
    caps = np.zeros((len(cap_st), GC_N))
    if len(cap_st):
        caps[:, 0] = (cap_st - st[0]) / (st[-1] - st[0]) * length
        caps[:, 1] = _per_station(mi, "cap_t", len(cap_st))
        if circ:
            caps[:, 2] = _per_station(mi, "cap_d_in", len(cap_st))
        else:
            caps[:, 2:4] = _per_station_pair(mi, "cap_d_in", len(cap_st))
    return gm, gs, caps


class MemberTable:
    """Members of one unit: ``members`` [nM,16], ``station_off`` [nM+1], ``stations`` [nSt,16], ``cap_off`` [nM+1],
    ``caps`` [nCap,4]."""

    def __init__(self, gms, gss, gcs=None):
        self.members = np.ascontiguousarray(np.array(gms, dtype=np.float64).reshape(-1, GM_N))
        counts = [len(g) for g in gss]
        self.station_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.stations = np.ascontiguousarray(np.concatenate(gss, axis=0) if gss else np.zeros((0, GS_N)))
        if gcs is None:
            gcs = [np.zeros((0, GC_N))] * len(gss)
        self.cap_off = np.concatenate([[0], np.cumsum([len(g) for g in gcs])]).astype(np.int64)
        self.caps = np.ascontiguousarray(np.concatenate(gcs, axis=0) if gcs else np.zeros((0, GC_N))).reshape(-1, GC_N)

    @property
    def n(self):
        return self.members.shape[0]


def describe_unit(design, heading_adjust=0.0, include_turbine=True):
    """MemberTable of a unit, in the order of FOWT.memberList (raft_fowt.py:223-272): every platform member once per
    heading, then towers and nacelle members.  The design dict is not modified."""
    if "joints" in design:
        raise UnsupportedMember("designs with explicit joints (multi-body / flexible units) are not generated on the device")
    plat = design["platform"]
    pmm = int(_scalar(plat, "potModMaster", default=0, kind=int))
    dls_default = _scalar(plat, "dlsMax", default=5.0)
    gms, gss, gcs = [], [], []
    for mi in plat["members"]:
        mi = dict(mi)
        if pmm in [1]:
            mi["potMod"] = False
        elif pmm in [2, 3]:
            mi["potMod"] = True
        if "dlsMax" not in mi:
            mi["dlsMax"] = dls_default
        headings = _as_given(mi, "heading", 0.)
        if np.isscalar(headings):
            headings = [headings]
        for h in headings:
            gm, gs, gc = describe_member(mi, heading=h + heading_adjust)
            gms.append(gm)
            gss.append(gs)
            gcs.append(gc)
    if include_turbine and "turbine" in design and design["turbine"] is not None:
        turb = design["turbine"]
        nrotors = int(_scalar(turb, "nrotors", default=1, kind=int))
        for key in ("tower", "nacelle"):
            if key in turb:
                items = turb[key]
                if isinstance(items, dict):
                    items = [items] * nrotors
                for mi in items:
                    gm, gs, gc = describe_member(dict(mi), part_of=key)
                    gms.append(gm)
                    gss.append(gs)
                    gcs.append(gc)
    return MemberTable(gms, gss, gcs)


class DesignTables:
    """Descriptors of a batch of designs, as raftx_build_designs takes them."""

    def __init__(self, member_off, members, station_off, stations, cap_off, caps):
        self.member_off, self.members = member_off, members
        self.station_off, self.stations = station_off, stations
        self.cap_off, self.caps = cap_off, caps

    @property
    def n_design(self):
        return len(self.member_off) - 1

    def __iter__(self):          # (memberOff, members, stationOff, stations): the four mandatory tables
        return iter((self.member_off, self.members, self.station_off, self.stations))

    def take(self, lo, hi):
        """Descriptors of designs [lo, hi): pure slices with re-based offsets."""
        m0, m1 = int(self.member_off[lo]), int(self.member_off[hi])
        s0, s1 = int(self.station_off[m0]), int(self.station_off[m1])
        c0, c1 = int(self.cap_off[m0]), int(self.cap_off[m1])
        return DesignTables(self.member_off[lo:hi + 1] - m0, self.members[m0:m1], self.station_off[m0:m1 + 1] - s0,
                            self.stations[s0:s1], self.cap_off[m0:m1 + 1] - c0, self.caps[c0:c1])


def _chain(offs):
    out = [np.zeros(1, dtype=np.int64)]
    base = 0
    for o in offs:
        out.append(o[1:] + base)
        base += o[-1]
    return np.concatenate(out).astype(np.int64)


def concat_units(tables):
    """DesignTables of a list of MemberTables (one per design)."""
    member_off = np.concatenate([[0], np.cumsum([t.n for t in tables])]).astype(np.int64)
    members = np.ascontiguousarray(np.concatenate([t.members for t in tables], axis=0))
    stations = np.ascontiguousarray(np.concatenate([t.stations for t in tables], axis=0))
    caps = np.ascontiguousarray(np.concatenate([t.caps for t in tables], axis=0))
    return DesignTables(member_off, members, _chain([t.station_off for t in tables]), stations,
                        _chain([t.cap_off for t in tables]), caps)


class SweepTables:
    """Descriptors of nDesign variants of ONE base unit (same members, stations and caps; different numbers), for
    parametric sweeps without a Python loop over designs: edit the arrays in place with NumPy broadcasting, then
    hand ``tables()`` to raftx_build_designs.

    ``members`` [nD,nM,16], ``stations`` [nD,nSt,16], ``caps`` [nD,nCap,4] start as copies of the base unit.
    ``set_ends`` moves a member's end points the way Member.__init__ does (heading rotation, length, station / cap /
    ballast positions rescaled with the length, raft_member.py:72-77,99,143,173)."""

    def __init__(self, base, n_design):
        self.base = base
        self.n = int(n_design)
        self.members = np.repeat(base.members[None], self.n, axis=0)
        self.stations = np.repeat(base.stations[None], self.n, axis=0)
        self.caps = np.repeat(base.caps[None], self.n, axis=0)
        length = base.members[:, GM_L]
        # positions as fractions of the member length (what the YAML's arbitrary station units mean)
        self._st_frac = base.stations[:, GS_S] / np.repeat(length, np.diff(base.station_off))
        self._fill_frac = base.stations[:, GS_LFILL] / np.repeat(length, np.diff(base.station_off))
        self._cap_frac = base.caps[:, 0] / np.repeat(length, np.diff(base.cap_off)) if len(base.caps) else np.zeros(0)

    def station_rows(self, m):
        return slice(int(self.base.station_off[m]), int(self.base.station_off[m + 1]))

    def cap_rows(self, m):
        return slice(int(self.base.cap_off[m]), int(self.base.cap_off[m + 1]))

    def set_ends(self, m, rA, rB, heading=0.0):
        """End points of member ``m`` for every design: rA, rB [nD,3] BEFORE the heading rotation [deg]."""
        rA = np.asarray(rA, dtype=float).reshape(self.n, 3)
        rB = np.asarray(rB, dtype=float).reshape(self.n, 3)
        length = np.sqrt(np.sum((rB - rA) ** 2, axis=1))
        if heading != 0.0:
            c, s = np.cos(np.deg2rad(heading)), np.sin(np.deg2rad(heading))
            rot = lambda r: np.stack([c * r[:, 0] + (-s) * r[:, 1], s * r[:, 0] + c * r[:, 1], r[:, 2]], axis=1)
            rA, rB = rot(rA), rot(rB)
        self.members[:, m, GM_RA:GM_RA + 3] = rA
        self.members[:, m, GM_RB:GM_RB + 3] = rB
        self.members[:, m, GM_L] = length
        rows = self.station_rows(m)
        self.stations[:, rows, GS_S] = self._st_frac[rows][None] * length[:, None]
        self.stations[:, rows, GS_LFILL] = self._fill_frac[rows][None] * length[:, None]
        crow = self.cap_rows(m)
        if crow.stop > crow.start:
            self.caps[:, crow, 0] = self._cap_frac[crow][None] * length[:, None]

    def set_diameter(self, m, d, d2=None):
        """Diameter (or side pair d, d2) of member ``m`` at all its stations: d [nD] or [nD,nStations]."""
        rows = self.station_rows(m)
        d = np.asarray(d, dtype=float)
        self.stations[:, rows, GS_D] = d[:, None] if d.ndim == 1 else d
        d2 = d if d2 is None else np.asarray(d2, dtype=float)
        self.stations[:, rows, GS_D + 1] = d2[:, None] if d2.ndim == 1 else d2

    def tables(self):
        nM, nSt, nCap = self.base.n, len(self.base.stations), len(self.base.caps)
        member_off = np.arange(self.n + 1, dtype=np.int64) * nM
        so = (self.base.station_off[None, :-1] + (np.arange(self.n, dtype=np.int64) * nSt)[:, None]).reshape(-1)
        co = (self.base.cap_off[None, :-1] + (np.arange(self.n, dtype=np.int64) * nCap)[:, None]).reshape(-1)
        return DesignTables(member_off, np.ascontiguousarray(self.members.reshape(-1, GM_N)),
                            np.concatenate([so, [self.n * nSt]]).astype(np.int64),
                            np.ascontiguousarray(self.stations.reshape(-1, GS_N)),
                            np.concatenate([co, [self.n * nCap]]).astype(np.int64),
                            np.ascontiguousarray(self.caps.reshape(-1, GC_N)))


# ---------------------------------------------------------------------- C3 workload (BASELINE configs[2])
def volturnus_sweep(base_design, scales, heading_adjust=0.0):
    """Member descriptors of the C3 sweep, vectorised over designs: the five parameters of
    raft/parametersweep.py:33-37 (centre-column d, outer-column d, draft, outer-column radius, pontoon height)
    times ``scales`` [nD,5], with the dependent-geometry edits of :56-87 -- applied straight to the
    descriptor arrays (no per-design Python).  ``base_design``: examples/VolturnUS-S_example.yaml (members: centre column,
    outer column x3, pontoon x3, upper beam x3, tower)."""
    scales = np.asarray(scales, dtype=float)
    nD = len(scales)
    base = describe_unit(base_design, heading_adjust=heading_adjust)
    heads = [np.atleast_1d(np.array(m.get("heading", 0.0), dtype=float)) for m in base_design["platform"]["members"]]
    assert [len(h) for h in heads] == [1, 3, 3, 3], "not the VolturnUS-S member layout"
    sw = SweepTables(base, nD)
    ccD, ocD, T, ocR, pH = 10.0 * scales[:, 0], 12.5 * scales[:, 1], -20.0 * scales[:, 2], 51.75 * scales[:, 3], 7.0 * scales[:, 4]
    z0 = np.zeros(nD)
    col = lambda *xs: np.stack([np.broadcast_to(np.asarray(x, dtype=float), (nD,)) for x in xs], axis=1)
    sw.set_ends(0, col(z0, z0, T), col(z0, z0, 15.0), heading=heads[0][0] + heading_adjust)
    sw.set_diameter(0, ccD)
    for c in range(3):
        h = heads[1][c] + heading_adjust
        sw.set_ends(1 + c, col(ocR, z0, T), col(ocR, z0, 15.0), heading=h)
        sw.set_diameter(1 + c, ocD)
        sw.set_ends(4 + c, col(ccD / 2, z0, T + pH / 2), col(ocR - ocD / 2, z0, T + pH / 2), heading=heads[2][c] + heading_adjust)
        sw.set_diameter(4 + c, np.full(nD, 12.4), pH)
        sw.set_ends(7 + c, col(ccD / 2, z0, 14.545), col(ocR - ocD / 2, z0, 14.545), heading=heads[3][c] + heading_adjust)
    return sw


class VariantProgram:
    """Edit program of a parametric sweep for the DEVICE (raftx_variant_program, include/raftx.h): the base unit's
    descriptors + edits that are affine in the sweep parameters -- what ``SweepTables.set_ends`` / ``set_diameter`` do in
    NumPy per batch (14 ms per 10 000 VolturnUS-S variants on one host thread, 66 MB over PCIe), stated once:

      ends(m, A, B, heading)   end points of member m BEFORE its heading rotation [deg]: A, B [3][nParam+1] rows of
                               coefficients (constant, then one per parameter) -- raft_member.py:41,72-77
      diameter(m, d, d2=None)  diameter (side pair) of every station of member m: [nParam+1] coefficients each

    The library evaluates value = c0 + c1*p1 + ... left to right without fused multiply-adds, takes the member length from
    the edited ends, rotates them by the heading and keeps station / ballast / cap positions at their fraction of the
    length -- bit for bit what ``SweepTables`` computes (tests/test_geometry.py)."""

    def __init__(self, base, n_param):
        self.base, self.n_param = base, int(n_param)
        nM, nSt = base.n, len(base.stations)
        self.end_coef = np.zeros((nM, 6, self.n_param + 1))
        self.end_edit = np.zeros(nM, dtype=np.int32)
        self.head_cs = np.tile([1.0, 0.0], (nM, 1))
        self.dia_coef = np.zeros((nSt, 2, self.n_param + 1))
        self.dia_edit = np.zeros(nSt, dtype=np.int32)

    def ends(self, m, A, B, heading=0.0):
        self.end_coef[m, :3] = np.asarray(A, dtype=float).reshape(3, self.n_param + 1)
        self.end_coef[m, 3:] = np.asarray(B, dtype=float).reshape(3, self.n_param + 1)
        self.end_edit[m] = 1
        self.head_cs[m] = [np.cos(np.deg2rad(heading)), np.sin(np.deg2rad(heading))] if heading != 0.0 else [1.0, 0.0]

    def diameter(self, m, d, d2=None):
        rows = slice(int(self.base.station_off[m]), int(self.base.station_off[m + 1]))
        d = np.asarray(d, dtype=float).reshape(self.n_param + 1)
        self.dia_coef[rows, 0] = d
        self.dia_coef[rows, 1] = d if d2 is None else np.asarray(d2, dtype=float).reshape(self.n_param + 1)
        self.dia_edit[rows] = 1

    def offsets(self, n_design):
        """(member_off, station_off, cap_off) of n_design variants: the uniform layout the library writes."""
        b = self.base
        nM, nSt, nCap = b.n, len(b.stations), len(b.caps)
        d = np.arange(n_design, dtype=np.int64)
        so = (b.station_off[None, :-1] + d[:, None] * nSt).reshape(-1)
        co = (b.cap_off[None, :-1] + d[:, None] * nCap).reshape(-1)
        return (np.arange(n_design + 1, dtype=np.int64) * nM, np.concatenate([so, [n_design * nSt]]).astype(np.int64),
                np.concatenate([co, [n_design * nCap]]).astype(np.int64))

    def tables(self, expanded, n_design):
        """DesignTables around descriptor arrays the library expanded (Context.expand_variants)."""
        gm, gs, gc = expanded
        mo, so, co = self.offsets(n_design)
        return DesignTables(mo, gm, so, gs, co, gc)


VOLTURNUS_PARAMS = ("centre-column diameter", "outer-column diameter", "draft (keel z)", "outer-column radius", "pontoon height")


def volturnus_params(scales):
    """Physical parameter values [nD,5] of the C3 sweep from the U[0.75,1.25] scale factors (raft/parametersweep.py:33-40)."""
    scales = np.asarray(scales, dtype=float)
    return np.ascontiguousarray(np.stack([10.0 * scales[:, 0], 12.5 * scales[:, 1], -20.0 * scales[:, 2], 51.75 * scales[:, 3],
                                          7.0 * scales[:, 4]], axis=1))


def volturnus_program(base_design, heading_adjust=0.0):
    """The dependent-geometry edits of raft/parametersweep.py:56-87 (as ``volturnus_sweep`` applies them) as a
    VariantProgram over the parameters (ccD, ocD, T, ocR, pH) = ``volturnus_params(scales)``."""
    base = describe_unit(base_design, heading_adjust=heading_adjust)
    heads = [np.atleast_1d(np.array(m.get("heading", 0.0), dtype=float)) for m in base_design["platform"]["members"]]
    assert [len(h) for h in heads] == [1, 3, 3, 3], "not the VolturnUS-S member layout"
    P = VariantProgram(base, 5)
    c = lambda c0=0.0, ccD=0.0, ocD=0.0, T=0.0, ocR=0.0, pH=0.0: [c0, ccD, ocD, T, ocR, pH]
    zero = c()
    P.ends(0, [zero, zero, c(T=1.0)], [zero, zero, c(15.0)], heading=heads[0][0] + heading_adjust)
    P.diameter(0, c(ccD=1.0))
    for k in range(3):
        P.ends(1 + k, [c(ocR=1.0), zero, c(T=1.0)], [c(ocR=1.0), zero, c(15.0)], heading=heads[1][k] + heading_adjust)
        P.diameter(1 + k, c(ocD=1.0))
        P.ends(4 + k, [c(ccD=0.5), zero, c(T=1.0, pH=0.5)], [c(ocD=-0.5, ocR=1.0), zero, c(T=1.0, pH=0.5)],
               heading=heads[2][k] + heading_adjust)
        P.diameter(4 + k, c(12.4), c(pH=1.0))
        P.ends(7 + k, [c(ccD=0.5), zero, c(14.545)], [c(ocD=-0.5, ocR=1.0), zero, c(14.545)], heading=heads[3][k] + heading_adjust)
    return P
