"""CPU affinity of a rank: bind the process to the cores of its GPU's NUMA node.

One process per GPU (the reference's `accelerate launch --multi_gpu`, run.sh:8).  On an 8-GPU MI355X node every GPU hangs off
one of two sockets; a rank whose staging copies and file reads run on the far socket pays the inter-socket hop for every H2D
of the database stream (search_tasks.py:107-116 feeds the model from a host memmap).  Linux only, sysfs only, fail-soft: any
missing piece returns None and leaves the affinity alone.
"""
from __future__ import annotations

import os
from typing import Optional


def _parse_cpulist(txt: str) -> set:
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_pci_address(dev_index: int) -> Optional[str]:
    """"dddd:bb:dd.f" of HIP device `dev_index` (torch device properties)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(dev_index)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


def numa_cpus_of_gpu(dev_index: int, sysfs: str = "/sys") -> Optional[tuple]:
    """-> (numa node, sorted cpu list) of the GPU's PCI device, or None."""
    addr = gpu_pci_address(dev_index)
    if addr is None:
        return None
    try:
        node = int(open(f"{sysfs}/bus/pci/devices/{addr}/numa_node").read())
        if node < 0:
            return None
        cpus = _parse_cpulist(open(f"{sysfs}/devices/system/node/node{node}/cpulist").read())
        return (node, sorted(cpus)) if cpus else None
    except Exception:
        return None


def bind_to_gpu_numa(dev_index: int) -> Optional[dict]:
    """Restrict this process to the CPUs of the GPU's NUMA node (intersected with what it is already allowed to use; ranks
    whose GPUs share a node share its cores -- a rank runs one driving thread plus the part-file writer's pool).  Returns a
    description or None (nothing changed)."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    info = numa_cpus_of_gpu(dev_index)
    if info is None:
        return None
    node, cpus = info
    try:
        allowed = sorted(set(cpus) & os.sched_getaffinity(0))
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return {"numa_node": node, "cpus": len(allowed), "first_cpu": allowed[0], "last_cpu": allowed[-1]}
    except Exception:
        return None
