"""Multi-GPU layer: reads / chunk ranges shard embarrassingly (one process per GPU, weights
replicated, no data-path collective); the only exchange is ONE all-reduce(sum) of the
per-label call counts at the end of a run — the distributed form of the label tally in
src/remora/validate.py:42-45 / get_label_counts src/remora/data_chunks.py:1074-1082.
`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI); "gloo" is used by the CPU tests.
The payload is int64[num_out] (<= 128 B): latency-bound, link bandwidth irrelevant."""
import os

import numpy as np


LAST_BINDING = None  # what bind_rank did for this process (setup_ranks / bench.py), for reports
_HOST_GROUP = None  # gloo side channel beside an nccl (= RCCL) default group: agreement on failures, and the fallback transport
_FALLBACK = None  # why the collectives of this process run over the gloo side channel instead of RCCL (None: they do not)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _single_forced():
    """REMORA_AMD_DIST_SINGLE=1: a world of ONE rank still builds its process group and sends every helper below through
    the backend's collectives (tests: the RCCL paths - device tensors, dtypes, object gathers - run on a 1-GPU box)."""
    return os.environ.get("REMORA_AMD_DIST_SINGLE") == "1"


def _collective():
    """True when the helpers below have a process group to talk to (more than one rank, or the forced single rank)."""
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _single_forced())


def _on_device():
    """True when the collectives of this process move DEVICE tensors through RCCL; False for gloo (tests, shared GPUs) and
    after `first_collective_ms` found that RCCL does not work on this node (every rank then uses the gloo side channel)."""
    import torch.distributed as dist

    return dist.get_backend() == "nccl" and _FALLBACK is None


def _group():
    """The process group the helpers talk to: the default one, or the gloo side channel after an RCCL failure."""
    return _HOST_GROUP if _FALLBACK is not None else None


def note_fallback(reason):
    """Record that this process's collectives run over gloo because RCCL could not be used (`reason`), for `transport()`."""
    global _FALLBACK
    _FALLBACK = str(reason)


def transport():
    """What carries this process's collectives, for reports: 'nccl', 'gloo', 'gloo (RCCL failed: ...)' or None."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None
    if _FALLBACK is not None:
        return f"gloo (RCCL failed: {_FALLBACK})"
    return str(dist.get_backend())


def shard_range(n, rank, world):
    """Contiguous [start, stop) of `n` units for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_process_group(backend=None, set_device=True, timeout_s=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1).  `timeout_s` bounds every
    collective of the group (a rank that never arrives becomes an exception, not a hang)."""
    import datetime

    import torch
    import torch.distributed as dist

    rank, world, local = env_rank_world()
    if world <= 1 and not _single_forced():
        return rank, world, local
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world <= 1 and "MASTER_PORT" not in os.environ:  # the forced single rank outside a launcher
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if backend == "nccl" and set_device:
            torch.cuda.set_device(local)
        kw = {"timeout": datetime.timedelta(seconds=float(timeout_s))} if timeout_s else {}
        if backend == "nccl":
            # a collective that cannot complete raises in the caller after the timeout (instead of the watchdog taking the
            # process down): first_collective_ms turns that into the gloo fallback
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        if os.environ.get("MASTER_ADDR") in ("127.0.0.1", "localhost") and os.path.isdir("/sys/class/net/lo"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: never resolve the container's hostname
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        global _HOST_GROUP
        if backend == "nccl" and world > 1 and _HOST_GROUP is None:
            try:  # rendezvous over the store only: no RCCL traffic
                _HOST_GROUP = dist.new_group(backend="gloo", **kw)
            except Exception as e:  # noqa: BLE001 - without it a failing RCCL is simply a failed run, as before
                import sys

                print(f"[remora_amd.dist] no gloo side channel: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                _HOST_GROUP = None
    return rank, world, local


def first_collective_ms():
    """Milliseconds the first collective of the process group takes (RCCL builds its communicator lazily, so this is the
    communicator's cost: bootstrap over TCP, topology search, xGMI ring setup); 0.0 for a single process."""
    import time

    import torch
    import torch.distributed as dist

    if not _collective():
        return 0.0
    global _FALLBACK
    t0 = time.perf_counter()
    err = None
    try:
        if os.environ.get("REMORA_AMD_DIST_FAIL_FIRST") == "1" and dist.get_backend() == "nccl":  # tests: the fallback below
            raise RuntimeError("REMORA_AMD_DIST_FAIL_FIRST")
        t = torch.ones(1, dtype=torch.int64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.synchronize()
        if int(t.item()) != dist.get_world_size():
            raise RuntimeError(f"first all-reduce returned {int(t.item())}, expected the world size {dist.get_world_size()}")
    except Exception as e:  # noqa: BLE001
        if _HOST_GROUP is None:
            raise
        err = f"{type(e).__name__}: {str(e).splitlines()[0][:200] if str(e) else ''}"
    if _HOST_GROUP is not None:
        # every rank learns whether RCCL worked for ALL of them: one failed rank moves the whole job onto the side channel
        ok = torch.tensor([0 if err else 1], dtype=torch.int64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=_HOST_GROUP)
        if int(ok.item()) == 0:
            reasons = [None] * dist.get_world_size()
            dist.all_gather_object(reasons, err, group=_HOST_GROUP)
            _FALLBACK = next((f"rank {i}: {r}" for i, r in enumerate(reasons) if r), "unknown")
            import sys

            if dist.get_rank() == 0:
                print(f"[remora_amd.dist] RCCL did not complete its first all-reduce ({_FALLBACK}); the collectives of this run "
                      f"(int64 label counts, clocks) go over gloo instead", file=sys.stderr, flush=True)
    return (time.perf_counter() - t0) * 1e3


def allgather_floats(xs):
    """[world][len(xs)] float64 numpy array of every rank's `xs` (diagnostics: per-rank timings)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(x) for x in xs], dtype=torch.float64)
    if not _collective():
        return t.numpy()[None, :].copy()
    if _on_device():
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t, group=_group())
    return torch.stack(out).cpu().numpy()


def launch_ranks(argv, n, scan_bam=None, command=None):
    """Start `n` ranks of `python -m remora_amd <argv>` on this node (one process per GPU or several, rendezvous on
    127.0.0.1) and return 0 when all of them did, else the first non-zero exit code (the other ranks are stopped).  Used by
    the CLI when `--gpus N` is given outside a launcher.  The ranks are started directly with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment - what torch.distributed.run would set, without the two
    seconds that launcher needs to import torch before it starts anything.  `scan_bam`: while the ranks start
    (interpreter, torch, model load) this process makes the one pass over that BAM which tells every rank where its
    share begins (io.write_bam_scan), instead of each rank inflating the whole file for itself.  `command` replaces
    `python -m remora_amd` (tests)."""
    import socket
    import subprocess
    import sys
    import tempfile
    import time

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    n = int(n)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # helper threads of a rank (OpenMP in torch, the batch gather): its share of the cores this process may use, between 2
    # and 8 - six ranks with eight each on a 16-core allowance cost 2-4 % (profiles/r03_infer_cli_threads_ab.log)
    from .util import effective_cpu_count

    per_rank = str(max(2, min(8, effective_cpu_count() // max(n, 1))))
    sized = [v for v in ("OMP_NUM_THREADS", "RMR_PACK_THREADS") if v not in env]
    for v in sized:
        env[v] = per_rank
    env["REMORA_AMD_LAUNCHER_SIZED"] = ",".join(sized)  # bind_rank may refine these, never a value the user gave
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    scan_path = None
    if scan_bam is not None:
        from .io import SCAN_ENV

        scan_path = os.path.join(tempfile.gettempdir(), f"remora_amd_scan_{os.getpid()}_{port}.npz")
        env[SCAN_ENV] = scan_path
    cmd = list(command or [sys.executable, "-m", "remora_amd"]) + list(argv)
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()

    try:
        if scan_path is not None:
            from .io import write_bam_scan

            write_bam_scan(scan_bam, scan_path)
        while True:  # a rank that fails takes the others with it (they would wait for it in the next collective)
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                stop_all()
                return bad[0]
            if all(c == 0 for c in codes):
                return 0
            time.sleep(0.02)
    except BaseException:
        stop_all()
        raise
    finally:
        if scan_path is not None and os.path.exists(scan_path):
            os.unlink(scan_path)


# ---- rank placement: a rank (its helper threads, its pinned staging rings) on the socket its GPU hangs off ----------------
# The reference has one reader process and nothing to place (src/remora/inference.py:488-519); here every rank feeds its own
# GPU from host memory (`infer` / `validate` / `dataset prepare` with --gpus N), so on a two-socket node a rank that runs on
# the far socket crosses the inter-socket link with every H2D byte.  Linux allocates first-touch on the node of the
# touching thread: binding the rank BEFORE it creates threads or pinned buffers places both.


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of sysfs cpulist files)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_pci_address(device):
    """'dddd:bb:dd.f' of torch device `device` (from the HIP device properties), or None when the build does not tell."""
    import torch

    try:
        p = torch.cuda.get_device_properties(int(device))
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:  # noqa: BLE001 - placement is best effort
        return None


def numa_of_pci(addr, sysfs="/sys"):
    """(numa node, its cpus) of the PCI device `addr`; (-1, []) when sysfs does not say (single socket, VM, container)."""
    if not addr:
        return -1, []
    base = os.path.join(sysfs, "bus", "pci", "devices", addr)
    try:
        node = int(open(os.path.join(base, "numa_node")).read().strip())
    except (OSError, ValueError):
        return -1, []
    cpus = []
    if node >= 0:
        try:
            cpus = parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
        except (OSError, ValueError):
            pass
    if not cpus:
        try:
            cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
        except (OSError, ValueError):
            cpus = []
    return node, cpus


def plan_rank_binding(gpu_addrs, procs_per_gpu=1, allowed_cpus=None, sysfs="/sys", cpu_budget=None):
    """Placement of every local rank of a `--gpus len(gpu_addrs) --procs-per-gpu P` run, as a pure function of the sysfs
    tree: [{rank, device, pci, numa_node, cpus, threads}] for local ranks 0 .. G*P-1 (device = rank // P).
    `cpus`: the allowed cores of the GPU's NUMA node (every allowed core when the node is unknown or none of its cores is
    allowed); `threads`: helper threads of the rank (OpenMP, the batch gather) = the cores its socket offers it - the
    node's allowed cores divided by the ranks placed on that node, also bounded by this process's CPU budget (cgroup
    quota) divided by all ranks - between 2 and 8."""
    allowed = sorted(set(allowed_cpus)) if allowed_cpus is not None else sorted(os.sched_getaffinity(0))
    n_ranks = len(gpu_addrs) * max(int(procs_per_gpu), 1)
    plan = []
    for r in range(n_ranks):
        dev = r // max(int(procs_per_gpu), 1)
        node, cpus = numa_of_pci(gpu_addrs[dev], sysfs)
        mine = [c for c in cpus if c in set(allowed)]
        if node < 0 or not mine:
            node, mine = (node if mine else -1), list(allowed)
        plan.append(dict(rank=r, device=dev, pci=gpu_addrs[dev], numa_node=node, cpus=mine))
    on_node = {}
    for p in plan:
        on_node[p["numa_node"]] = on_node.get(p["numa_node"], 0) + 1
    budget = int(cpu_budget) if cpu_budget else len(allowed)
    for p in plan:
        share = min(len(p["cpus"]) // on_node[p["numa_node"]], max(budget // max(n_ranks, 1), 1))
        p["threads"] = max(2, min(8, share))
    return plan


def bind_rank(device, local_rank=None, local_world=None, procs_per_gpu=1, sysfs="/sys", gpu_addrs=None):
    """Bind THIS process to the cores of the NUMA node of its GPU and size its helper-thread pools for its share of that
    socket; call it before any thread or pinned buffer exists.  Returns the rank's entry of plan_rank_binding plus
    `bound` (whether sched_setaffinity was applied).  REMORA_AMD_RANK_BINDING=0 turns it off; never raises.
    `gpu_addrs` replaces the PCI addresses asked of the HIP runtime (tests)."""
    info = dict(rank=local_rank, device=device, pci=None, numa_node=-1, cpus=[], threads=None, bound=False)
    if os.environ.get("REMORA_AMD_RANK_BINDING", "1") == "0":
        return info
    try:
        import torch

        from .util import effective_cpu_count

        local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else int(local_world)
        n_gpus = max(local_world // max(int(procs_per_gpu), 1), 1)
        forced = os.environ.get("REMORA_AMD_FORCE_DEVICE")
        addrs = gpu_addrs if gpu_addrs is not None else [
            gpu_pci_address(int(forced) if forced is not None else d) if (forced is not None or d < torch.cuda.device_count())
            else None for d in range(n_gpus)]
        plan = plan_rank_binding(addrs, procs_per_gpu, sysfs=sysfs, cpu_budget=effective_cpu_count())
        me = dict(plan[min(local_rank, len(plan) - 1)])
        me["bound"] = False
        if me["numa_node"] >= 0 and me["cpus"]:
            os.sched_setaffinity(0, me["cpus"])
            # threads that already exist (the HIP runtime's, started by the device query above) do not inherit: bind them too
            try:
                for tid in os.listdir("/proc/self/task"):
                    try:
                        os.sched_setaffinity(int(tid), me["cpus"])
                    except (OSError, ValueError):
                        pass
            except OSError:
                pass
            me["bound"] = True
        # the thread pools read these when they are created / per call.  A value the USER set stays (launch_ranks marks the
        # ones it sized itself, without the topology, in REMORA_AMD_LAUNCHER_SIZED: those are replaced by the plan's)
        sized = set(os.environ.get("REMORA_AMD_LAUNCHER_SIZED", "").split(","))
        if "TORCHELASTIC_RUN_ID" in os.environ and os.environ.get("OMP_NUM_THREADS") == "1":
            sized.add("OMP_NUM_THREADS")  # torch.distributed.run's own default for a variable nobody set
        for var in ("OMP_NUM_THREADS", "RMR_PACK_THREADS"):
            if var not in os.environ or var in sized:
                os.environ[var] = str(me["threads"])
        try:
            torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
        except Exception:  # noqa: BLE001
            pass
        return me
    except Exception as e:  # noqa: BLE001 - placement must never cost a run
        info["error"] = f"{type(e).__name__}: {e}"
        return info


def setup_ranks(gpus, procs_per_gpu=1, backend=None, timeout_s=600.0):
    """(rank, world, device) of this process for a `--gpus N [--procs-per-gpu P]` command line: (0, 1, None) for a single
    process; under a launcher the process group is created and the device is LOCAL_RANK // P.  P > 1 puts several
    processes on each GPU - the host side of a file-to-file run (record parsing, per-read arithmetic, tag formatting,
    BGZF) is Python and scales with processes, as the reference's reader / prepare workers do
    (src/remora/inference.py:488-572); their kernels share the GPU.  Transport of the one small collective: RCCL when
    every rank owns a GPU, gloo when GPUs are shared (RCCL refuses two ranks on one device) or when
    REMORA_AMD_DIST_BACKEND / `backend` says so.  REMORA_AMD_FORCE_DEVICE pins every rank to one GPU (tests).  A world
    that differs from gpus x P is an error, never a silently smaller run."""
    rank, world, local = env_rank_world()
    gpus, procs_per_gpu = int(gpus), max(int(procs_per_gpu), 1)
    if gpus * procs_per_gpu <= 1 and world <= 1:
        return 0, 1, None
    if world != gpus * procs_per_gpu:
        from . import RemoraError

        raise RemoraError(f"--gpus {gpus} x --procs-per-gpu {procs_per_gpu} but the launcher started WORLD_SIZE={world} rank(s)")
    forced = os.environ.get("REMORA_AMD_FORCE_DEVICE")
    backend = backend or os.environ.get("REMORA_AMD_DIST_BACKEND") or ("gloo" if procs_per_gpu > 1 or forced is not None else None)
    device = int(forced) if forced is not None else local // procs_per_gpu
    import torch

    if torch.cuda.is_available():  # whatever the transport: everything that resolves "the current device" must land on this rank's GPU
        torch.cuda.set_device(device)
        global LAST_BINDING
        LAST_BINDING = bind_rank(device, local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), procs_per_gpu)
        if os.environ.get("RMR_INFER_TIMING"):
            import sys

            b = LAST_BINDING
            print(f"[rank {rank}/{world}] cuda:{device} pci {b.get('pci')} numa node {b.get('numa_node')} bound {b.get('bound')} "
                  f"cores {len(b.get('cpus') or [])} helper threads {b.get('threads')}", file=sys.stderr, flush=True)
    init_process_group(backend, set_device=False, timeout_s=timeout_s)
    return rank, world, device


def barrier():
    import torch.distributed as dist

    if _collective():
        dist.barrier(group=_group())


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (small picklable bookkeeping: per-reason read counts, file
    names); [obj] for a single process.  Not a data-path collective."""
    import torch.distributed as dist

    if not _collective():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj, group=_group())
    return out


def gather_arrays(arr):
    """Every rank's numpy array (same dtype and trailing shape, any length) concatenated in rank order, on every rank."""
    import torch
    import torch.distributed as dist

    arr = np.ascontiguousarray(arr)
    if not _collective():
        return arr
    on_gpu = _on_device()
    n = torch.tensor([arr.shape[0]], dtype=torch.int64)
    n = n.cuda() if on_gpu else n
    sizes = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, n, group=_group())
    sizes = [int(x.item()) for x in sizes]
    width = max(sizes)
    pad = np.zeros((width,) + arr.shape[1:], arr.dtype)
    pad[: arr.shape[0]] = arr
    t = torch.from_numpy(pad)
    t = t.cuda() if on_gpu else t
    parts = [torch.empty_like(t) for _ in sizes]
    dist.all_gather(parts, t, group=_group())
    return np.concatenate([p.cpu().numpy()[:k] for p, k in zip(parts, sizes)], axis=0)


def allreduce_counts(counts):
    """Sum per-label counts over all ranks.  `counts`: int64 torch tensor (on the GPU for
    nccl/RCCL, CPU for gloo) or numpy array; returns the same type, reduced in place for tensors."""
    import torch
    import torch.distributed as dist

    if not _collective():
        return counts
    if isinstance(counts, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(counts, np.int64))
        if _on_device():
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_group())
        return t.cpu().numpy()
    if counts.is_cuda and not _on_device():  # gloo (tests, shared GPUs, RCCL fallback): reduce through the host
        t = counts.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_group())
        counts.copy_(t)
        return counts
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=_group())
    return counts


def allreduce_max_float(x):
    """max over ranks of a python float (used for the max-over-ranks step time)."""
    import torch
    import torch.distributed as dist

    if not _collective():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64)
    if _on_device():
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_group())
    return float(t.item())


def allgather_counts(counts):
    """Per-rank label counts as an int64 numpy array [world, num_out] (rank order) on every rank; one row for a
    single process.  Used to check the all-reduce: the rows must add up to the reduced counts."""
    import torch
    import torch.distributed as dist

    t = counts if hasattr(counts, "is_cuda") else torch.from_numpy(np.ascontiguousarray(counts, np.int64))
    if not _collective():
        return t.detach().cpu().numpy()[None, :].copy()
    if _on_device():
        t = t.cuda() if not t.is_cuda else t
    else:
        t = t.cpu()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t.contiguous(), group=_group())
    return torch.stack(out).cpu().numpy()


# ---- the C-ABI collective (include/remora_hip.h: rmr_comm_unique_id / rmr_comm_init / rmr_allreduce_counts) -----------
# RCCL called from inside libremora_hip.so: what a non-torch host (the C/C++ caller of INTEGRATION.md) uses.  The
# 128-byte unique id travels from rank 0 to the other ranks by whatever channel the job has; here torch.distributed.
def init_cabi_comm(engine, rank=None, world=None):
    """Create the library's own RCCL communicator on `engine` (collective over all ranks).  Returns True when a
    communicator exists afterwards (False for a single process without torch.distributed, where none is needed)."""
    import ctypes

    import torch
    import torch.distributed as dist

    from . import _lib as L

    lib = L.lib()
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
    ident = (ctypes.c_uint8 * 128)()
    if rank == 0:
        L.check(lib.rmr_comm_unique_id(ident))
    if world > 1:
        t = torch.tensor(list(ident), dtype=torch.uint8)
        if _on_device():
            t = t.to(engine.torch_device)  # the rank's own GPU (the current device is per thread)
        dist.broadcast(t, src=0, group=_group())
        ident = (ctypes.c_uint8 * 128)(*t.cpu().tolist())
    L.check(lib.rmr_comm_init(engine.handle, ident, int(rank), int(world)))
    return True


def cabi_allreduce_counts(engine, counts):
    """All-reduce(sum) of an int64 device tensor / numpy array through the library's RCCL communicator, in place."""
    from . import _lib as L

    lib = L.lib()
    if isinstance(counts, np.ndarray):
        assert counts.dtype == np.int64 and counts.flags.c_contiguous
        L.check(lib.rmr_allreduce_counts(engine.handle, counts.ctypes.data, counts.size, L.MEM_HOST))
        return counts
    assert counts.is_cuda and counts.dtype.is_floating_point is False and counts.element_size() == 8
    L.check(lib.rmr_allreduce_counts(engine.handle, counts.data_ptr(), counts.numel(), L.MEM_DEVICE))
    engine.synchronize()
    return counts


def cabi_allreduce_check(engine, per_rank, rank, world, timeout_s=60.0):
    """bench.py's cross-check of the C-ABI collective against torch.distributed's result on the same counts: every rank
    contributes its own row of `per_rank`; the reduced vector must equal the column sums.  Runs in a worker thread with a
    deadline (a collective that cannot complete must not hang the bench line); never raises."""
    import threading

    import torch
    import torch.distributed as dist

    if world > 1 and (not dist.is_initialized() or not _on_device()):
        return {"status": "skipped", "reason": f"ranks do not talk RCCL (transport: {transport()})"}
    res = {}

    def work():
        try:
            torch.cuda.set_device(engine.torch_device)  # a new thread starts on device 0
            init_cabi_comm(engine, rank, world)
            mine = torch.from_numpy(np.ascontiguousarray(per_rank[rank], np.int64)).to(engine.torch_device)
            cabi_allreduce_counts(engine, mine)
            got = mine.cpu().numpy()
            want = np.sum(per_rank, axis=0)
            res.update(status="ok" if np.array_equal(got, want) else "mismatch", got=[int(x) for x in got], world=world,
                       via="rmr_comm_unique_id / rmr_comm_init / rmr_allreduce_counts (RCCL inside libremora_hip.so)")
            from . import _lib as L

            L.check(L.lib().rmr_comm_destroy(engine.handle))
        except Exception as e:  # noqa: BLE001
            res.update(status="error", error=f"{type(e).__name__}: {e}")

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return {"status": "timeout", "timeout_s": timeout_s}
    return res
