"""Duck-typed stand-ins for the reference's Model / FOWT / Member objects (now in raft_amd/snapshot.py; this module
keeps the names the tests grew up with)."""
from raft_amd.snapshot import *          # noqa: F401,F403
from raft_amd.snapshot import GOLDEN_DIR, MEMBER_ARRAYS, FOWT_ARRAYS, FOWT_SCALARS, Obj, _flatten, _unflatten   # noqa: F401
