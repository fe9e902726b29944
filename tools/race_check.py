#!/usr/bin/env python3
"""Static race check of the multi-stream schedule of the reduced-system Cholesky, without a GPU.

Under tools/hipstub (dry-run HIP runtime, kernels do not run) the library issues exactly the launches, event records and
stream waits it would issue on the device; the stub keeps them in issue order with their streams and, for the kernels of
the factorisation, their arguments (block column k, pointer to the tile list, count -- host memory under the stub).  From
that this module builds the happens-before relation HIP guarantees (FIFO order inside a stream; work enqueued after
hipStreamWaitEvent(s, e) follows everything that preceded the captured hipEventRecord(e)) as vector clocks, derives the set
of 128x128 tiles every launch reads and writes

    k_panel128(S, k, rows)            read+write (k,k) and (I,k) for the listed row tiles I
    k_syrk<KT>(S, k, pairs (I,J), n)  read+write (I,J); read (I,k..k+KT-1), (J,k..k+KT-1)

and checks that every two launches touching a common tile, at least one of them writing, are ordered (two updates of the
same tile are read-modify-write: they must be ordered too, and the order is what makes the sums reproducible).  Every other
operation (the kernels that build S before and the solves after, memsets, copies) is treated as touching everything: it
must be ordered against all launches of the factorisation -- that is the fork / join of the schedule.

Must run inside the LD_PRELOAD=libhipstub.so process (it dereferences the list pointers):  tests/test_schedule_races.py.
"""
import ctypes as C
import re

OP_LAUNCH, OP_RECORD, OP_WAIT, OP_MEMOP, OP_SYNC = 1, 2, 3, 4, 5


class Op(C.Structure):
    _fields_ = [("type", C.c_int), ("grid", C.c_uint), ("stream", C.c_void_p), ("obj", C.c_void_p), ("args", C.c_ulonglong * 8)]


def bind(stub):
    stub.hipstub_trace_ops.restype = C.POINTER(Op)
    stub.hipstub_kernel_name.restype = C.c_char_p
    stub.hipstub_kernel_name.argtypes = [C.c_void_p]
    return stub


def trace(stub):
    """-> list of dicts in issue order."""
    n = stub.hipstub_trace_count(); ops = stub.hipstub_trace_ops()
    out = []
    for i in range(n):
        o = ops[i]
        d = {"type": o.type, "stream": o.stream or 0, "obj": o.obj or 0, "grid": o.grid}
        if o.type == OP_LAUNCH:
            d["name"] = stub.hipstub_kernel_name(o.obj).decode()
            d["args"] = [int(a) for a in o.args[:4]]
        out.append(d)
    return out


def _i32_list(ptr, n):
    return list((C.c_int32 * n).from_address(ptr)) if n > 0 else []


def accesses(op):
    """(reads, writes) tile sets of a launch of the factorisation, or None for any other operation."""
    name = op.get("name", "")
    if "k_panel128" in name:
        k = op["args"][1] & 0xFFFFFFFF
        rows = _i32_list(op["args"][2], (op["grid"] - 1) // 2)
        t = {(k, k)} | {(int(i), k) for i in rows}
        return t, t
    m = re.search(r"k_syrkILi(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        kt = int(m.group(1)); k0 = op["args"][1] & 0xFFFFFFFF; npairs = op["args"][3] & 0xFFFFFFFF
        flat = _i32_list(op["args"][2], 2 * npairs)
        reads, writes = set(), set()
        for q in range(npairs):
            i, j = int(flat[2 * q]), int(flat[2 * q + 1])
            writes.add((i, j))
            for kk in range(k0, k0 + kt):
                reads.add((i, kk)); reads.add((j, kk))
        return reads | writes, writes
    return None


def check(ops, max_report=10):
    """Vector-clock happens-before + conflict check.  Returns a summary dict; `races` lists unordered conflicting pairs."""
    streams = {}
    for o in ops:
        if o["type"] in (OP_LAUNCH, OP_RECORD, OP_WAIT, OP_MEMOP):
            streams.setdefault(o["stream"], len(streams))
    ns = len(streams)
    clock = {s: [0] * ns for s in streams}       # clock of the last operation enqueued in the stream (incl. inherited waits)
    event_clock = {}
    nodes = []                                    # (index in ops, stream index, vector clock, reads, writes | None)
    for idx, o in enumerate(ops):
        t = o["type"]
        if t == OP_RECORD:
            event_clock[o["obj"]] = list(clock[o["stream"]])
        elif t == OP_WAIT:
            ec = event_clock.get(o["obj"])
            if ec is not None:                    # waiting on a never-recorded event is a no-op in HIP
                c = clock[o["stream"]]
                clock[o["stream"]] = [max(a, b) for a, b in zip(c, ec)]
        elif t in (OP_LAUNCH, OP_MEMOP):
            si = streams[o["stream"]]
            c = list(clock[o["stream"]]); c[si] += 1
            clock[o["stream"]] = c
            acc = accesses(o) if t == OP_LAUNCH else None
            nodes.append((idx, si, c, acc))

    def hb(a, b):      # a happens-before b  (a was issued first)
        return a[2][a[1]] <= b[2][a[1]]
    races, checks = [], 0
    last_writer, readers = {}, {}                # per tile
    since_global, last_global = [], None
    n_fact = 0
    for nd in nodes:
        acc = nd[3]
        if acc is None:                           # touches everything: after all launches since the previous such operation
            for x in since_global:
                checks += 1
                if not hb(x, nd):
                    races.append(("join", x[0], nd[0]))
            since_global = []; last_global = nd
            last_writer.clear(); readers.clear()
            continue
        n_fact += 1
        if last_global is not None:
            checks += 1
            if not hb(last_global, nd):
                races.append(("fork", last_global[0], nd[0]))
        reads, writes = acc
        for tl in reads:
            w = last_writer.get(tl)
            if w is not None and w is not nd:
                checks += 1
                if not hb(w, nd):
                    races.append(("read-after-write", w[0], nd[0], tl))
        for tl in writes:
            for r in readers.get(tl, ()):
                if r is not nd:
                    checks += 1
                    if not hb(r, nd):
                        races.append(("write-after-read", r[0], nd[0], tl))
        for tl in writes:
            last_writer[tl] = nd; readers[tl] = []
        for tl in reads - writes:
            readers.setdefault(tl, []).append(nd)
        since_global.append(nd)
    desc = []
    for r in races[:max_report]:
        a, b = ops[r[1]], ops[r[2]]
        desc.append({"kind": r[0], "first": a.get("name", "memop")[:60], "first_k": (a.get("args") or [0, 0, 0])[2] & 0xFFFFFFFF,
                     "second": b.get("name", "memop")[:60], "second_k": (b.get("args") or [0, 0, 0])[2] & 0xFFFFFFFF,
                     "tile": list(r[3]) if len(r) > 3 else None})
    return {"operations": len(ops), "launches": sum(1 for o in ops if o["type"] == OP_LAUNCH), "factorisation_launches": n_fact,
            "streams": ns, "ordered_conflicts_checked": checks, "races": len(races), "first_races": desc}
