"""CPU / memory affinity of one rank of a one-node launch (tools/run_node.sh).

    python tools/rank_affinity.py LOCAL_RANK N_RANKS     ->  prints  "<cpu list> <numa node>"   (node -1 = unknown)

The NUMA node of rank r is the one of HIP device r: PCI address from the HIP runtime (torch.cuda.get_device_properties), then
/sys/bus/pci/devices/<address>/numa_node -- NOT the glob order of /sys/class/drm/card* (card10 sorts before card2, and other
AMD functions sit there too: ADVICE r3).  The node's cpulist is SLICED between the ranks whose GPUs hang off the same node, so
that 8 ranks with their PNG pools do not share cores; without NUMA information the cores are split evenly by rank."""
import os
import sys


def parse_cpulist(s):
    out = []
    for part in s.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


def fmt_cpulist(cpus):
    return ','.join(str(c) for c in cpus)


def gpu_numa_nodes(n):
    """NUMA node per HIP device index (None where unknown)."""
    nodes = [None] * n
    try:
        import torch
        for r in range(min(n, torch.cuda.device_count())):
            p = torch.cuda.get_device_properties(r)
            addr = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id)
            path = '/sys/bus/pci/devices/%s/numa_node' % addr
            if os.path.exists(path):
                v = int(open(path).read().strip())
                nodes[r] = v if v >= 0 else None
    except Exception:
        pass
    return nodes


def affinity(rank, n, nodes, node_cpus, all_cpus):
    """(cpus, node) of `rank`: its slice of its NUMA node's cores, else an even split of all cores."""
    node = nodes[rank] if rank < len(nodes) else None
    if node is not None and node_cpus.get(node):
        mates = [r for r in range(n) if r < len(nodes) and nodes[r] == node]
        cpus = node_cpus[node]
        per = max(1, len(cpus) // len(mates))
        i = mates.index(rank)
        return cpus[i * per:(i + 1) * per] or cpus, node
    per = max(1, len(all_cpus) // n)
    return all_cpus[rank * per:(rank + 1) * per] or all_cpus, -1


def main():
    rank, n = int(sys.argv[1]), int(sys.argv[2])
    all_cpus = sorted(os.sched_getaffinity(0))
    nodes = gpu_numa_nodes(n)
    node_cpus = {}
    for nd in {x for x in nodes if x is not None}:
        path = '/sys/devices/system/node/node%d/cpulist' % nd
        if os.path.exists(path):
            node_cpus[nd] = [c for c in parse_cpulist(open(path).read()) if c in set(all_cpus)]
    cpus, node = affinity(rank, n, nodes, node_cpus, all_cpus)
    print(fmt_cpulist(cpus), node)


if __name__ == '__main__':
    main()
