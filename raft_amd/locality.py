"""Host placement of a rank: run the feeding thread on the socket its GPU hangs off.

One process per GPU streams ~16 GB/s of member descriptors out of page-locked memory.  On a two-socket
host a rank whose pages live on the far socket pushes that stream over the socket link and shares it with its
neighbours; pinning the process to the GPU's NUMA node BEFORE the ctx and its landing areas are allocated keeps the
first-touch pages local.  Everything here is best effort: unknown topology (containers without sysfs, single-socket
hosts, node -1) leaves the process where it was and says so in the returned record.
"""
import os


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def node_cpus(node, sysfs="/sys/devices/system/node"):
    """CPUs of a NUMA node from sysfs, or an empty set."""
    try:
        with open(os.path.join(sysfs, "node%d" % node, "cpulist")) as f:
            return _parse_cpulist(f.read())
    except (OSError, ValueError):
        return set()


def bind_near_device(rlib, device_id, sysfs="/sys/devices/system/node"):
    """Restrict this process to the CPUs of the NUMA node of ``device_id``.  Returns a record
    {"pci": ..., "numa_node": ..., "cpus": n or None, "bound": bool, "why": ...} for the bench line / logs."""
    rec = {"pci": "", "numa_node": -1, "cpus": None, "bound": False, "why": ""}
    try:
        rec["pci"], rec["numa_node"] = rlib.device_locality(device_id)
    except Exception as e:                              # the library could not tell: stay put
        rec["why"] = "device_locality: %s" % e
        return rec
    if rec["numa_node"] < 0:
        rec["why"] = "sysfs reports no NUMA node for the device"
        return rec
    cpus = node_cpus(rec["numa_node"], sysfs)
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set()
    target = cpus & allowed
    if not target:
        rec["why"] = "no allowed CPU on node %d" % rec["numa_node"]
        return rec
    if target == allowed:
        rec.update(cpus=len(target), why="already confined to the node")
        return rec
    try:
        os.sched_setaffinity(0, target)
    except OSError as e:
        rec["why"] = "sched_setaffinity: %s" % e
        return rec
    rec.update(cpus=len(target), bound=True)
    return rec
