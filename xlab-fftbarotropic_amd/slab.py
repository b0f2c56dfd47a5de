"""Slab-decomposed RK4 stepping over the GPUs of one node.

One process per GPU.  Physical fields are split by x rows, spectral fields by ky columns: the ACTIVE columns
(ky < world*KA: every mode inside the dealiasing circle) evenly over the ranks, the FROZEN columns beyond them likewise
(their modes are masked, fftwfop.cpp:57-61, and never change, SURVEY.md note N1).  Every 2-D transform needs one
all-to-all transpose between its row pass and its column pass: per RK stage one of the four derivative fields
(columns -> rows) and one of the tendency (rows -> columns), active columns only.  The reference has no counterpart (it
is single-process); the decomposition computes exactly what main.cpp:286-317 computes.

Two drivers with one schedule:
  * `EngineSlab` -- the product: the whole step (local HIP passes, exchange buffers, two streams, RCCL grouped
    send/recv) lives behind the C ABI (`fb_slab_*`, csrc/fb_slab_driver.h); this class only bootstraps the transport
    (RCCL unique id over torch.distributed; the in-process hub; a gloo callback for rehearsals on one GPU).
  * `SlabModel(backend=...)` -- the same schedule spelled in Python over a pluggable compute backend, so that the
    exchange logic (who sends which block to whom, in which order) runs on CPU under gloo with the numpy test double of
    tests/slab_numpy_backend.py.  `stage_schedule()` is checked against the engine's `fb_slab_plan`.
There is no fallback from one to the other.
"""
import ctypes as C
import math

import numpy as np

OP_COL_BWD, OP_XCHG_W4, OP_ROW, OP_XCHG_T, OP_COL_FWD, OP_COL_ALL_BWD = 1, 2, 3, 4, 5, 6


def _round16(v):
    return (v + 15) // 16 * 16


def slab_geometry(nx, ny, world):
    """(XL rows, KA active columns, KF frozen columns) per rank -- must match slab_split() in csrc/fftbaro.hip
    (tests/test_slab_cpu.py compares with fb_slab_geometry)."""
    hy = ny // 2 + 1
    dxw, dyw = math.ceil(float(np.float32(nx)) / 3.0), math.ceil(float(np.float32(ny)) / 3.0)
    gws = float(np.float32(float(dxw) ** 2 + float(dyw) ** 2))                 # fftwfop.cpp:57
    jmax = 0
    while jmax < hy and float(jmax) * float(jmax) < gws:
        jmax += 1
    ka = _round16((jmax + world - 1) // world)
    nf = hy - world * ka
    kf = _round16((nf + world - 1) // world) if nf > 0 else 0
    return nx // world, ka, kf


def slab_col_groups(nx, ny, world):
    """Columns per rank of the active column groups (one, or two where a stage is pipelined by column groups) -- mirrors
    slab_active_groups() / fb_slab_col_groups in the engine.  A rank's active slab [rank*KA, (rank+1)*KA) is cut locally: its
    first n_0 columns are group 0, the rest group 1."""
    import os
    _, ka, _ = slab_geometry(nx, ny, world)
    na = 1
    if world > 1 and ka >= 32:
        if 17.5 * nx * ka * 8.0 / 5e6 >= 100.0:              # one rank's column work of a stage, microseconds at ~5 TB/s
            na = 2
        if os.environ.get("FB_SLAB_COL_GROUPS") in ("1", "2"):
            na = int(os.environ["FB_SLAB_COL_GROUPS"])
    tiles = ka // 16
    return [16 * (tiles // na + (1 if g < tiles % na else 0)) for g in range(na)]


def stage_plan(nx, ny, world):
    """(field groups, row chunks) of one RK stage's two transposes -- mirrors slab_plan() in csrc/fb_slab_driver.h."""
    xl, ka, _ = slab_geometry(nx, ny, world)
    bwd_us, row_us = 2.0 * nx * ka * 8.0 / 5e6, 5.0 * xl * (ny // 2 + 1) * 8.0 / 4e6      # one field's backward sub-pass, the row pass
    fg = 1 if world == 1 else (4 if bwd_us >= 20.0 else (2 if bwd_us >= 10.0 else 1))
    ch = 1 if world == 1 else (2 if row_us >= 100.0 else 1)
    while ch > 1 and ((xl // ch) & 1 or xl % ch):
        ch >>= 1
    if len(slab_col_groups(nx, ny, world)) > 1:
        fg = 1                                               # pipelined by column groups: a group's four fields leave together
    return fg, ch


def stage_schedule(nx, ny, world):
    """Operations of one RK stage in issue order, as (kind, argument) -- mirrors fb_slab_plan."""
    fg, ch = stage_plan(nx, ny, world)
    ops = []
    ncg = len(slab_col_groups(nx, ny, world))
    if ncg > 1:                                              # arguments of OP_XCHG_W4 / OP_COL_*: the column group
        for h in range(ch):
            ops += [(OP_ROW, h), (OP_XCHG_T, h)]
        for g in range(ncg):
            ops += [(OP_COL_FWD, g), (OP_COL_ALL_BWD, g), (OP_XCHG_W4, g)]
        return ops
    for g in range(fg):
        ops.append((OP_COL_BWD, g))
        if world > 1:
            ops.append((OP_XCHG_W4, g))
    for h in range(ch):
        ops.append((OP_ROW, h))
        if world > 1:
            ops.append((OP_XCHG_T, h))
    ops.append((OP_COL_FWD, 0))
    return ops


LINK_GBS = 60.0         # assumed xGMI rate per direction and peer (7 links x ~153 GB/s bidirectional per GPU: <= 77 GB/s one way)
GROUP_LATENCY_MS = 0.03  # assumed cost of one grouped ncclSend/ncclRecv beyond its bytes


def predicted_step_ms(nx, ny, world, local_ms_per_step=None):
    """DESIGN.md section 6's model of one multi-GPU RK4 step, as numbers (a PREDICTION: no node has been available to the builder).
    Per stage a rank sends 5 fields x XL x KA complex to every peer, all peers at once on their own links:
        t_link   = 5 XL KA 8 B / LINK_GBS  +  2 dependent collectives x GROUP_LATENCY_MS
        local    = the rank's passes of a stage: local_ms_per_step / 4 where measured (bench.py's null-transport run), else
                   22.4 C / world at 5 TB/s
        exposed  = the part of `local` no transfer hides: with two column groups the first row chunk (5.0/22.5 of local / chunks; the
                   shares are the kernels' counter bytes per stage, DESIGN.md section 4: 8.3 + 1.9 + 7.3 + 5.0 C); otherwise the forward
                   x pass + update (10.2/22.5), the first field group's backward sub-pass (7.3/22.5 / groups) and the first row chunk
        stage    = t_link + exposed;  step = 4 stages.
    Returns (ms per step, dict of the terms)."""
    xl, ka, _ = slab_geometry(nx, ny, world)
    fg, ch = stage_plan(nx, ny, world)
    ncg = len(slab_col_groups(nx, ny, world))
    hy = ny // 2 + 1
    if local_ms_per_step is None:
        local = 22.4 * 8.0 * nx * _round16(hy) / world / 5e12 * 1e3
    else:
        local = local_ms_per_step / 4.0
    if world == 1:
        return 4.0 * local, {"t_link_ms": 0.0, "local_ms": local, "exposed_ms": local}
    t_link = 5.0 * xl * ka * 8.0 / (LINK_GBS * 1e9) * 1e3 + 2 * GROUP_LATENCY_MS
    if ncg > 1:
        exposed = local * (5.0 / 22.5) / ch
    else:
        exposed = local * (10.2 / 22.5 + 7.3 / 22.5 / fg + 5.0 / 22.5 / ch)
    return 4.0 * (t_link + exposed), {"t_link_ms_per_stage": t_link, "local_ms_per_stage": local, "exposed_ms_per_stage": exposed,
                                      "link_GBs_assumed": LINK_GBS, "group_latency_ms_assumed": GROUP_LATENCY_MS,
                                      "local_from": "measured (null transport)" if local_ms_per_step is not None else "22.4 C / world at 5 TB/s"}


def local_rows(field, rank, world):
    xl = field.shape[0] // world
    return np.ascontiguousarray(field[rank * xl:(rank + 1) * xl])


# ---------------------------------------------------------------------------------------------------------------
# the product driver: everything behind the C ABI
# ---------------------------------------------------------------------------------------------------------------
class _DevMem:
    """A raw device allocation of the engine, viewed by torch through __cuda_array_interface__."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class EngineSlab:
    """One rank of the engine-driven multi-GPU model (fb_slab_*).  transport:
       "rccl"  -- ncclCommInitRank with an id that rank 0 creates and torch.distributed broadcasts (the product path)
       "gloo"  -- torch.distributed point-to-point behind the callback transport (ranks may share one GPU: rehearsal)
       hub     -- an integer handle from `local_hub(world)`: all ranks are threads of this process (rehearsal)
       "null"  -- exchanges move nothing (wrong fields, right timing of the local passes)
       None    -- world == 1"""

    def __init__(self, nx, ny=None, Lx=600000.0, Ly=600000.0, nu=6.5, dt=3.0, rank=0, world=1, transport=None, dist=None):
        import torch
        from . import binding as B
        ny = ny or nx
        self.torch, self.B, self.L = torch, B, B.lib()
        self.nx, self.ny, self.rank, self.world, self.dist = nx, ny, rank, world, dist
        h = C.c_void_p()
        B.check(self.L.fb_slab_create(C.byref(h), nx, ny, Lx, Ly, nu, dt, rank, world))
        self._h = h
        v = [C.c_int() for _ in range(7)]
        B.check(self.L.fb_slab_info(self._h, *[C.byref(x) for x in v]))
        self.XL, self.KA, self.KF, self.kyA0, self.kyF0, self.field_groups, self.row_chunks = [x.value for x in v]
        assert world == 1 or (self.XL, self.KA, self.KF) == slab_geometry(nx, ny, world)      # one rank: one group of all columns at the engine's pitch
        ng, cols = C.c_int(), (C.c_int * 2)()
        B.check(self.L.fb_slab_col_groups(nx, ny, world, C.byref(ng), cols))
        self.col_groups = [cols[g] for g in range(ng.value)]                                  # 2 entries: the stage is pipelined by column groups
        self._cb = None
        self.transport = "none"
        if world > 1:
            if transport == "rccl":
                # All ranks or none: ncclCommInitRank is collective, so one rank that quietly took another transport would leave the
                # others hanging in it.  The outcome of every step is agreed on over the process group; any failure raises on EVERY
                # rank (there is no automatic second choice -- ask for transport="gloo" explicitly to rehearse without RCCL).
                self._connect_rccl()
                self.transport = "rccl (engine: grouped ncclSend/ncclRecv)"
            elif transport == "gloo":
                self._connect_gloo()
                self.transport = "torch.distributed %s point-to-point (callback)" % dist.get_backend()
            elif isinstance(transport, int):
                B.check(self.L.fb_slab_connect_local(self._h, C.c_void_p(transport)))
                self.transport = "local (threads of one process)"
            elif transport == "null":
                # moves nothing: the fields are garbage, the local passes and the schedule are the real ones -- bench.py times it
                # to report how much of a multi-GPU step is local work
                self._cb = self.B.ALLTOALL_FN(lambda user, send, recv, stride, offset, count, stream: 0)
                B.check(self.L.fb_slab_connect_callback(self._h, self._cb, None))
                self.transport = "null (timing of the local passes only)"
            else:
                raise B.FftBaroError("EngineSlab: world > 1 needs transport='rccl', 'gloo' or a local hub handle")

    # -- transports
    def _agree(self, ok, what):
        """all_reduce(MIN) of a success flag over the process group: every rank learns whether ALL ranks got through `what`."""
        torch, dist = self.torch, self.dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            raise self.B.FftBaroError("EngineSlab rank %d: %s failed on %s -- no rank connects" %
                                      (self.rank, what, "this rank" if not ok else "another rank"))

    def _connect_rccl(self):
        torch, dist = self.torch, self.dist
        idbuf = C.create_string_buffer(128)
        err = None
        if self.rank == 0:
            try:
                self.B.check(self.L.fb_slab_unique_id(idbuf))
            except self.B.FftBaroError as e:                  # rank 0 still takes part in the broadcast (a zero id) and in the vote below
                err = e
                idbuf = C.create_string_buffer(128)
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(list(idbuf.raw), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        if err is not None:
            import sys
            print("EngineSlab rank 0: cannot create the RCCL unique id: %s" % err, file=sys.stderr)
        self._agree(err is None, "ncclGetUniqueId")
        raw = bytes(t.cpu().tolist())
        try:
            self.B.check(self.L.fb_slab_connect_rccl(self._h, C.create_string_buffer(raw, 128)))
        except self.B.FftBaroError as e:
            import sys
            print("EngineSlab rank %d: ncclCommInitRank failed: %s" % (self.rank, e), file=sys.stderr)
            err = e
        self._agree(err is None, "ncclCommInitRank")

    def _connect_gloo(self):
        torch, dist, world, rank = self.torch, self.dist, self.world, self.rank

        def alltoall(user, send, recv, stride, offset, count, stream):
            try:
                torch.cuda.synchronize()                                     # the engine's streams are not torch's
                ops, keep = [], []
                for p in range(world):
                    s = torch.as_tensor(_DevMem(send + 4 * (p * stride + offset), count), device="cuda")
                    r = torch.as_tensor(_DevMem(recv + 4 * (p * stride + offset), count), device="cuda")
                    if p == rank:
                        r.copy_(s)
                    else:
                        keep += [s, r]
                        ops.append(dist.P2POp(dist.isend, s, p))
                        ops.append(dist.P2POp(dist.irecv, r, p))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                torch.cuda.synchronize()
                return 0
            except Exception as e:                                            # never unwind through the C frame
                import sys
                print("slab gloo transport failed: %r" % (e,), file=sys.stderr)
                return 1
        self._cb = self.B.ALLTOALL_FN(alltoall)                               # keep the trampoline alive
        self.B.check(self.L.fb_slab_connect_callback(self._h, self._cb, None))

    # -- model surface (this rank's rows)
    def _rows(self, a):
        t = self.torch
        if isinstance(a, np.ndarray):
            a = t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
        assert a.is_cuda and a.dtype == t.float32 and a.is_contiguous() and tuple(a.shape) == (self.XL, self.ny)
        # the engine reads the buffer on ITS streams: whatever torch still has queued for it (a fill, a copy) must have landed
        t.cuda.current_stream().synchronize()
        return a

    def set_vort_local(self, rows):
        a = self._rows(rows)
        self.B.check(self.L.fb_slab_set_vort_local(self._h, C.c_void_p(a.data_ptr())))
        self.synchronize()

    def set_source_local(self, rows):
        if rows is None:
            self.B.check(self.L.fb_slab_set_source_local(self._h, None))
        else:
            a = self._rows(rows)
            self.B.check(self.L.fb_slab_set_source_local(self._h, C.c_void_p(a.data_ptr())))
            self.synchronize()

    def step(self, n=1):
        self.B.check(self.L.fb_slab_step(self._h, n))

    def vort_local(self):
        out = self.torch.empty((self.XL, self.ny), dtype=self.torch.float32, device="cuda")
        self.B.check(self.L.fb_slab_get_vort_local(self._h, C.c_void_p(out.data_ptr())))
        self.synchronize()
        return out

    def diag_local(self):
        """This rank's rows of psi, u, v (the stage-0 record dumps, main.cpp:181-222)."""
        t = self.torch
        psi, u, v = (t.empty((self.XL, self.ny), dtype=t.float32, device="cuda") for _ in range(3))
        self.B.check(self.L.fb_slab_get_diag_local(self._h, C.c_void_p(psi.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(v.data_ptr())))
        self.synchronize()
        return psi, u, v

    def transport_selftest(self, count=1 << 18):
        """A known pattern of world*count floats through the connected transport; returns the number of wrong words (0 = links fine).
        Collective: every rank calls it."""
        bad = C.c_size_t()
        self.B.check(self.L.fb_slab_transport_selftest(self._h, count, C.byref(bad)))
        return int(bad.value)

    def transport_info(self):
        """What is connected, as the transport's own communicator reports it (fb_slab_transport_info): for RCCL the values of
        ncclCommCount / ncclCommUserRank / ncclCommCuDevice -- the proof that `world` ranks joined ONE communicator -- and this
        rank's HIP device ordinal.  -1 where the transport has no communicator (gloo callback, in-process hub)."""
        name = C.create_string_buffer(32)
        v = [C.c_int() for _ in range(4)]
        self.B.check(self.L.fb_slab_transport_info(self._h, name, 32, *[C.byref(x) for x in v]))
        return {"name": name.value.decode(), "comm_ranks": v[0].value, "comm_rank": v[1].value, "comm_device": v[2].value,
                "hip_device": v[3].value}

    def time_steps(self, n):
        ms = C.c_float()
        self.B.check(self.L.fb_slab_time_steps(self._h, n, C.byref(ms)))
        return ms.value

    def synchronize(self):
        self.B.check(self.L.fb_slab_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self.L.fb_slab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def local_hub(world):
    """Handle of an in-process rendezvous for `world` EngineSlab ranks driven by `world` threads (one GPU)."""
    from . import binding as B
    h = C.c_void_p()
    B.check(B.lib().fb_local_hub_create(C.byref(h), world))
    return h.value


def local_hub_destroy(hub):
    from . import binding as B
    B.lib().fb_local_hub_destroy(C.c_void_p(hub))


def engine_plan(nx, ny, world):
    """(field groups, row chunks, [(kind, argument), ...]) as the engine reports them (fb_slab_plan; no GPU needed)."""
    from . import binding as B
    fg, ch = C.c_int(), C.c_int()
    ops = (C.c_int * 64)()
    n = B.lib().fb_slab_plan(nx, ny, world, C.byref(fg), C.byref(ch), ops, 64)
    return fg.value, ch.value, [(ops[i] // 16, ops[i] % 16) for i in range(n)]


# ---------------------------------------------------------------------------------------------------------------
# the same schedule in Python over a pluggable compute backend (CPU rehearsal of the exchange logic)
# ---------------------------------------------------------------------------------------------------------------
def _all_to_all(dist, recv, send, world, stride, offset, count):
    """Block (offset, count) of every peer's stride-sized slot: send[p*stride + offset ...] -> peer p's recv[me*stride + offset ...]."""
    rank = dist.get_rank()
    ops = []
    for p in range(world):
        s, r = send[p * stride + offset:p * stride + offset + count], recv[p * stride + offset:p * stride + offset + count]
        if p == rank:
            r.copy_(s)
        else:
            ops.append(dist.P2POp(dist.isend, s, p))
            ops.append(dist.P2POp(dist.irecv, r, p))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


class SlabModel:
    """RK4 driver on `world` ranks, schedule in Python.  With backend=None this is the engine (EngineSlab)."""

    def __new__(cls, nx, ny=None, Lx=600000.0, Ly=600000.0, nu=6.5, dt=3.0, rank=0, world=1, backend=None, dist=None, transport=None):
        if backend is None:                                   # the product: everything behind the C ABI
            if dist is None and world > 1:
                import torch.distributed as dist
            if transport is None and world > 1:
                transport = "rccl" if dist.get_backend() == "nccl" else "gloo"
            return EngineSlab(nx, ny, Lx, Ly, nu, dt, rank, world, transport, dist)
        return super().__new__(cls)

    def __init__(self, nx, ny=None, Lx=600000.0, Ly=600000.0, nu=6.5, dt=3.0, rank=0, world=1, backend=None, dist=None, transport=None):
        ny = ny or nx
        self.nx, self.ny, self.rank, self.world = nx, ny, rank, world
        if dist is None and world > 1:
            import torch.distributed as dist
        self.dist = dist
        self.be = backend
        self.XL, self.KA, self.KF = slab_geometry(nx, ny, world)
        assert (self.be.XL, self.be.KA, self.be.KF) == (self.XL, self.KA, self.KF)
        self.field_groups, self.row_chunks = stage_plan(nx, ny, world)
        self.cols = slab_col_groups(nx, ny, world) + ([self.KF] if self.KF else [])     # columns per rank of every group, the engine's order
        self.nact = len(self.cols) - (1 if self.KF else 0)
        assert list(self.be.ncols) == self.cols
        self.primed = 0

    def _xchg(self, recv, send, stride, offset, count):
        if self.world > 1:
            _all_to_all(self.dist, recv, send, self.world, stride, offset, count)

    # -- state
    def set_vort_local(self, vort_rows):
        be = self.be
        assert tuple(vort_rows.shape) == (self.XL, self.ny)
        be.r2c_rows(vort_rows)                                                    # -> t_send (every group)
        for g, n in enumerate(self.cols):
            self._xchg(be.t_recv[g], be.t_send[g], self.XL * n, 0, self.XL * n)
        be.r2c_cols()
        self.primed = 0

    def set_source_local(self, src_rows):
        self.be.set_source(src_rows)

    def vort_local(self):
        be = self.be
        be.c2r_cols()                                                             # -> t_recv, [dst][XL][ncols]
        for g, n in enumerate(self.cols):
            self._xchg(be.t_send[g], be.t_recv[g], self.XL * n, 0, self.XL * n)
        return be.c2r_rows()

    def step(self, n=1):
        be = self.be
        if n <= 0:
            return
        if not self.primed:
            be.prime()                                                            # derivatives of every column; frozen ones final
            if self.KF:
                gf = self.nact
                self._xchg(be.w4_recv[gf], be.w4_send[gf], 4 * self.XL * self.KF, 0, 4 * self.XL * self.KF)
            self.primed = 1
        if self.nact > 1 and self.primed == 1:                                    # slab_groups_prologue
            for g in range(self.nact):
                be.col_bwd(0, 4, g)
                self._xchg(be.w4_recv[g], be.w4_send[g], 4 * self.XL * self.cols[g], 0, 4 * self.XL * self.cols[g])
            self.primed = 2
        rows = self.XL // self.row_chunks
        for _ in range(n):
            for k in range(4):                                                    # main.cpp:288-317
                for kind, arg in stage_schedule(self.nx, self.ny, self.world):
                    if self.nact > 1:                                             # pipelined by column groups: arg of the column operations = the group
                        if kind == OP_ROW:
                            be.row(arg * rows, rows)
                        elif kind == OP_XCHG_T:
                            for g in range(self.nact):
                                nc = self.cols[g]
                                self._xchg(be.t_recv[g], be.t_send[g], self.XL * nc, arg * rows * nc, rows * nc)
                        elif kind == OP_COL_FWD:
                            be.col_fwd(k, arg)
                        elif kind == OP_COL_ALL_BWD:
                            be.col_bwd(0, 4, arg)
                        else:
                            self._xchg(be.w4_recv[arg], be.w4_send[arg], 4 * self.XL * self.cols[arg], 0, 4 * self.XL * self.cols[arg])
                        continue
                    fld = self.XL * self.cols[0]
                    if kind == OP_COL_BWD:
                        be.col_bwd(4 * arg // self.field_groups, 4 * (arg + 1) // self.field_groups, 0)
                    elif kind == OP_XCHG_W4:
                        f0, f1 = 4 * arg // self.field_groups, 4 * (arg + 1) // self.field_groups
                        self._xchg(be.w4_recv[0], be.w4_send[0], 4 * fld, f0 * fld, (f1 - f0) * fld)
                    elif kind == OP_ROW:
                        be.row(arg * rows, rows)
                    elif kind == OP_XCHG_T:
                        self._xchg(be.t_recv[0], be.t_send[0], fld, arg * rows * self.cols[0], rows * self.cols[0])
                    else:
                        be.col_fwd(k, 0)

    def close(self):
        self.be.close()
