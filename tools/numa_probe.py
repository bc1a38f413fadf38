"""Where do pinned host buffers land, and what does that do to the PCIe copy rate?  (round 3, streaming regime)

For every NUMA node whose CPUs this process may run on: bind the thread to that node's CPUs, allocate and
first-touch pinned buffers there (hipHostMalloc), report the node the pages really sit on (move_pages
query), then time host->device and device->host copies from / to them, alone and both directions at once.
Also prints the GPU's own NUMA node (sysfs) and the process's allowed CPUs / memory nodes."""
import ctypes as C, glob, json, os, re, sys, time

hip = C.CDLL("libamdhip64.so")
libc = C.CDLL(None, use_errno=True)
H2D, D2H = 1, 2


def ck(rc, what):
    if rc:
        raise RuntimeError(f"{what} -> hip error {rc}")


def parse_list(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def page_nodes(ptr, nbytes, max_pages=64):
    """node of (a sample of) the pages of [ptr, ptr + nbytes): move_pages(2) with nodes = NULL"""
    page = 4096
    npages = min(max_pages, nbytes // page)
    stride = max(1, (nbytes // page) // npages)
    pages = (C.c_void_p * npages)(*[ptr + i * stride * page for i in range(npages)])
    status = (C.c_int * npages)()
    rc = libc.syscall(279, 0, C.c_ulong(npages), pages, None, status, 0)   # __NR_move_pages (x86_64)
    if rc != 0:
        return {"error": os.strerror(C.get_errno())}
    hist = {}
    for s in status:
        hist[int(s)] = hist.get(int(s), 0) + 1
    return hist


def rate(fn, nbytes, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


def main():
    info = {}
    buf = C.create_string_buffer(64)
    ck(hip.hipDeviceGetPCIBusId(buf, 64, 0), "hipDeviceGetPCIBusId")   # the host's other GPUs are in sysfs too
    d = "/sys/bus/pci/devices/" + buf.value.decode().lower()
    info["gpu_sysfs"] = d
    for f in ("numa_node", "local_cpulist", "current_link_speed", "current_link_width"):
        try:
            info["gpu_" + f] = open(f"{d}/{f}").read().strip()
        except OSError as e:
            info["gpu_" + f] = str(e)
    st = open("/proc/self/status").read()
    info["cpus_allowed"] = re.search(r"Cpus_allowed_list:\s*(\S+)", st).group(1)
    info["mems_allowed"] = re.search(r"Mems_allowed_list:\s*(\S+)", st).group(1)
    allowed = set(os.sched_getaffinity(0))
    nodes = {}
    for nd in sorted(glob.glob("/sys/devices/system/node/node*")):
        k = int(nd.rsplit("node", 1)[1])
        cpus = set(parse_list(open(nd + "/cpulist").read())) & allowed
        nodes[k] = sorted(cpus)
    info["node_cpus_allowed"] = {k: (f"{v[0]}..{v[-1]} ({len(v)})" if v else "none") for k, v in nodes.items()}
    ck(hip.hipSetDevice(0), "hipSetDevice")
    nbytes = 12_000_000
    dbuf = C.c_void_p()
    ck(hip.hipMalloc(C.byref(dbuf), C.c_size_t(2 * nbytes)), "hipMalloc")
    s_up, s_dn = C.c_void_p(), C.c_void_p()
    ck(hip.hipStreamCreateWithFlags(C.byref(s_up), 1), "stream")
    ck(hip.hipStreamCreateWithFlags(C.byref(s_dn), 1), "stream")
    results = []
    for k, cpus in nodes.items():
        if not cpus:
            continue
        os.sched_setaffinity(0, cpus)
        time.sleep(0.01)
        up, dn = C.c_void_p(), C.c_void_p()
        ck(hip.hipHostMalloc(C.byref(up), C.c_size_t(nbytes), 0), "hipHostMalloc")
        ck(hip.hipHostMalloc(C.byref(dn), C.c_size_t(nbytes), 0), "hipHostMalloc")
        C.memset(up, 1, nbytes)
        C.memset(dn, 2, nbytes)
        r = {"alloc_on_node_cpus": k, "pages_up": page_nodes(up.value, nbytes), "pages_dn": page_nodes(dn.value, nbytes)}

        def f_up():
            ck(hip.hipMemcpyAsync(dbuf, up, C.c_size_t(nbytes), H2D, s_up), "h2d")
            ck(hip.hipStreamSynchronize(s_up), "sync")

        def f_dn():
            ck(hip.hipMemcpyAsync(dn, C.c_void_p(dbuf.value + nbytes), C.c_size_t(nbytes), D2H, s_dn), "d2h")
            ck(hip.hipStreamSynchronize(s_dn), "sync")

        def f_both():
            ck(hip.hipMemcpyAsync(dbuf, up, C.c_size_t(nbytes), H2D, s_up), "h2d")
            ck(hip.hipMemcpyAsync(dn, C.c_void_p(dbuf.value + nbytes), C.c_size_t(nbytes), D2H, s_dn), "d2h")
            ck(hip.hipStreamSynchronize(s_up), "sync")
            ck(hip.hipStreamSynchronize(s_dn), "sync")

        def f_up_q8():   # eight copies in flight on the stream, one wait: the streaming loop's shape
            for _ in range(8):
                ck(hip.hipMemcpyAsync(dbuf, up, C.c_size_t(nbytes), H2D, s_up), "h2d")
            ck(hip.hipStreamSynchronize(s_up), "sync")

        for name, fn, nb in (("h2d_GBps", f_up, nbytes), ("d2h_GBps", f_dn, nbytes), ("both_GBps_total", f_both, 2 * nbytes),
                             ("h2d_q8_GBps", f_up_q8, 8 * nbytes)):
            for _ in range(20):
                fn()   # clocks
            r[name] = round(rate(fn, nb, 40), 2)
        results.append(r)
        hip.hipHostFree(up)
        hip.hipHostFree(dn)
    os.sched_setaffinity(0, allowed)
    info["per_node"] = results
    print(json.dumps(info, indent=1))


if __name__ == "__main__":
    main()
