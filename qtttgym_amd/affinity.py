"""Bind a rank to the CPU cores local to its GPU (one process per GPU; at 262 144 boards per GPU a step
is host-paced, 3.5 - 4.5 us per launch, so a Python loop that wanders over the sockets is the first suspect for a
sub-linear 8-GPU result).

No torch, no HIP: everything is read from sysfs, so it can run before anything initialises the GPU —
/sys/class/kfd/kfd/topology/nodes/*/properties (the order HIP enumerates the GPUs in, before *_VISIBLE_DEVICES) gives each
GPU's PCI address, /sys/bus/pci/devices/<address>/{numa_node,local_cpulist} the cores next to it.  Anything unreadable,
a node of -1 with no narrower core list, or a core list outside the process's allowed set = do nothing and say so.
Where the KFD topology is not readable (the containers of this pool: PermissionError) bind_to_pci() does the same from the
device's own PCI address, which the caller gets from the runtime once it is up (hipDeviceGetPCIBusId /
torch.cuda.get_device_properties): what matters is where the LAUNCHING thread runs, and that can be set at any time.
On the pool's two-socket hosts (256 CPUs, four GPUs per socket) the binding itself works (128 cores of the GPU's node); it
did NOT remove the run-to-run spread of launch-bound batches there (262 144 boards: 3.4 - 4.7 us per launch bound and unbound
alike, profiles/r05/bench_socket_binding_ab.txt) — it is kept for the 8-rank case, where eight unpinned Python loops share a host.
"""
import os

KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"
PCI_DEVICES = "/sys/bus/pci/devices"


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0,1,2,3,8,10,11}"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def format_cpulist(cpus):
    out, run = [], []
    for c in sorted(cpus):
        if run and c == run[-1] + 1:
            run.append(c)
            continue
        if run:
            out.append("%d-%d" % (run[0], run[-1]) if len(run) > 1 else "%d" % run[0])
        run = [c]
    if run:
        out.append("%d-%d" % (run[0], run[-1]) if len(run) > 1 else "%d" % run[0])
    return ",".join(out)


def kfd_gpus(root=KFD_NODES):
    """PCI addresses of the GPUs in KFD node order: ['0000:05:00.0', ...]"""
    gpus = []
    for name in sorted(os.listdir(root), key=int):
        props = {}
        with open(os.path.join(root, name, "properties")) as f:
            for line in f:
                k, _, v = line.strip().partition(" ")
                props[k] = v
        if int(props.get("simd_count", "0")) == 0:
            continue                                    # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        gpus.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return gpus


def _hip_level(environ):
    """HIP_VISIBLE_DEVICES and CUDA_VISIBLE_DEVICES are ONE remapping level of the HIP runtime, not two: the CUDA_ name is
    an alias that is read only when the HIP_ name is unset (launchers commonly export both with the same list — applying
    both would map a permutation twice and pin a rank to another GPU's cores)."""
    v = environ.get("HIP_VISIBLE_DEVICES")
    if v is None or v.strip() == "":
        v = environ.get("CUDA_VISIBLE_DEVICES")
    return v


def _levels(environ):
    """the remapping levels in effect, inner (HIP runtime) -> outer (ROCr), as strings"""
    return [v for v in (_hip_level(environ), environ.get("ROCR_VISIBLE_DEVICES")) if v is not None and v.strip() != ""]


def visible_index(local_index, environ=None):
    """the KFD-order index behind HIP device `local_index`, through HIP_ (or, when that is unset, CUDA_) VISIBLE_DEVICES and
    then ROCR_VISIBLE_DEVICES when they are plain integer lists (a UUID list is not resolved: None)"""
    environ = os.environ if environ is None else environ
    idx = local_index
    for v in _levels(environ):                                                               # inner -> outer
        try:
            ids = [int(x) for x in v.split(",")]
        except ValueError:
            return None
        if idx >= len(ids):
            return None
        idx = ids[idx]
    return idx


def visible_count(n_kfd, environ=None):
    """how many devices HIP should enumerate given n_kfd GPUs in the topology and the *_VISIBLE_DEVICES lists (None if a
    list is not plain integers)"""
    environ = os.environ if environ is None else environ
    n = n_kfd
    for v in reversed(_levels(environ)):                                                     # outer -> inner
        try:
            ids = [int(x) for x in v.split(",")]
        except ValueError:
            return None
        n = len([i for i in ids if 0 <= i < n])
    return n


def bind_to_pci(address, pci_root=PCI_DEVICES, apply=True):
    """Restrict this process to the cores local to the PCI device `address` ('0000:72:00.0').  Same result dict as
    bind_to_gpu, 'via': 'pci'.  Never raises."""
    info = {"bound": False, "numa_node": None, "cpus": None, "cpulist": None, "reason": None, "via": "pci", "device": address}
    try:
        allowed = os.sched_getaffinity(0)
        info["cpus"], info["cpulist"] = len(allowed), format_cpulist(allowed)
        dev = os.path.join(pci_root, address)
        with open(os.path.join(dev, "numa_node")) as f:
            info["numa_node"] = int(f.read().strip())
        with open(os.path.join(dev, "local_cpulist")) as f:
            local = parse_cpulist(f.read())
        want = local & allowed
        if not want:
            info["reason"] = "the GPU's local cores are outside this process's allowed set"
        elif want == allowed:
            info["reason"] = "the allowed set is already the GPU's local cores"
        elif apply:
            os.sched_setaffinity(0, want)
            info.update(bound=True, cpus=len(want), cpulist=format_cpulist(want))
        else:
            info.update(cpus=len(want), cpulist=format_cpulist(want), reason="apply=False")
    except (OSError, ValueError, AttributeError) as e:
        info["reason"] = "%s: %s" % (type(e).__name__, e)
    return info


def bind_to_gpu(local_index, kfd_root=KFD_NODES, pci_root=PCI_DEVICES, apply=True, expected_devices=None):
    """Restrict this process to the cores local to HIP device `local_index`.  Returns a small dict for the bench line:
    {'bound': bool, 'numa_node': int | None, 'cpus': n, 'cpulist': '...', 'reason': why not}.  Never raises.
    expected_devices: the number of devices HIP enumerates (torch.cuda.device_count(), which does not initialise the GPU):
    if the topology (through the *_VISIBLE_DEVICES lists) does not account for exactly that many — a container that hides
    GPUs by other means — the index -> PCI address mapping is not known and nothing is bound."""
    info = {"bound": False, "numa_node": None, "cpus": None, "cpulist": None, "reason": None}
    try:
        allowed = os.sched_getaffinity(0)
        info["cpus"], info["cpulist"] = len(allowed), format_cpulist(allowed)
        k = visible_index(int(local_index))
        gpus = kfd_gpus(kfd_root)
        if expected_devices is not None and visible_count(len(gpus)) != int(expected_devices):
            info["reason"] = ("the topology lists %d GPUs (%s through the *_VISIBLE_DEVICES lists), HIP enumerates %d: "
                              "which PCI device is device %s is not known" % (len(gpus), visible_count(len(gpus)),
                                                                              int(expected_devices), local_index))
            return info
        if k is None or k >= len(gpus):
            info["reason"] = "device %s not found in the KFD topology (%d GPUs)" % (local_index, len(gpus))
            return info
        dev = os.path.join(pci_root, gpus[k])
        with open(os.path.join(dev, "numa_node")) as f:
            info["numa_node"] = int(f.read().strip())
        with open(os.path.join(dev, "local_cpulist")) as f:
            local = parse_cpulist(f.read())
        want = local & allowed
        if not want:
            info["reason"] = "the GPU's local cores are outside this process's allowed set"
        elif want == allowed:
            info["reason"] = "the allowed set is already the GPU's local cores"
        elif apply:
            os.sched_setaffinity(0, want)
            info.update(bound=True, cpus=len(want), cpulist=format_cpulist(want))
        else:
            info.update(cpus=len(want), cpulist=format_cpulist(want), reason="apply=False")
    except (OSError, ValueError, AttributeError) as e:
        info["reason"] = "%s: %s" % (type(e).__name__, e)
    return info
