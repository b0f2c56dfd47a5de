"""Slab-decomposed RK4 stepping over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  Physical fields are
split by x rows, spectral fields by ky columns; every 2-D transform needs one all-to-all
transpose between its row pass and its column pass: per RK stage one transpose of the four
derivative fields (columns -> rows) and one of the tendency (rows -> columns).  The local
passes are the engine's HIP kernels behind `fb_model_phase` (include/fftbaro.h); this module
owns the four exchange buffers and the collectives.  The reference has no counterpart (it is
single-process); the decomposition reproduces main.cpp:286-317 exactly as the fused single-GPU
path does.

The compute backend is pluggable only so that the exchange logic can be exercised on CPU with
`gloo` (tests/test_slab_cpu.py supplies a numpy backend built on the oracle); the product
backend is `HipBackend` and there is no fallback to any other.
"""
import ctypes as C

import numpy as np

PH_PRIME, PH_COL_BWD, PH_ROW, PH_COL_FWD, PH_R2C_ROWS, PH_R2C_COLS, PH_C2R_COLS, PH_C2R_ROWS = range(8)


def slab_geometry(nx, ny, world):
    """(XL rows per rank, KS columns per slab) -- must match fb_create_slab (fftbaro.hip)."""
    hy = ny // 2 + 1
    ks = (hy + 16 * world - 1) // (16 * world) * 16
    return nx // world, ks


def local_rows(field, rank, world):
    xl = field.shape[0] // world
    return np.ascontiguousarray(field[rank * xl:(rank + 1) * xl])


class HipBackend:
    """Local passes on this rank's GPU through the C ABI."""

    def __init__(self, nx, ny, Lx, Ly, nu, dt, rank, world):
        import torch
        from . import binding as B
        self.torch, self.B = torch, B
        self.nx, self.ny, self.rank, self.world = nx, ny, rank, world
        self.L = B.lib()
        h = C.c_void_p()
        B.check(self.L.fb_create_slab(C.byref(h), nx, ny, Lx, Ly, rank, world))
        self.ctx = h
        B.check(self.L.fb_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        xl, ks, ky0, e = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
        B.check(self.L.fb_slab_geometry(self.ctx, C.byref(xl), C.byref(ks), C.byref(ky0), C.byref(e)))
        self.XL, self.KS, self.E = xl.value, ks.value, e.value
        if world > 1:                               # one GPU: the engine may have autotuned a larger pitch
            assert (self.XL, self.KS) == slab_geometry(nx, ny, world)
        # exchange buffers as flat float32 (re,im interleaved): the dtype every RCCL collective takes
        self.FL = 2 * self.E                    # tensor elements per field
        z = lambda n: torch.zeros(n, dtype=torch.float32, device="cuda")
        self.w4_send, self.w4_recv = z(4 * self.FL), z(4 * self.FL)
        if world > 1:
            self.t_send, self.t_recv = z(self.FL), z(self.FL)
        else:                                   # no exchange: the passes hand over in place
            self.w4_recv = self.w4_send
            self.t_send = self.t_recv = z(self.FL)
        m = C.c_void_p()
        B.check(self.L.fb_model_create_slab(C.byref(m), self.ctx, nu, dt, self.w4_send.data_ptr(), self.w4_recv.data_ptr(),
                                            self.t_send.data_ptr(), self.t_recv.data_ptr()))
        self.model = m

    def phase(self, ph, stage=0, real_in=None, real_out=None):
        self.B.check(self.L.fb_model_phase(self.model, ph, stage,
                                           C.c_void_p(real_in.data_ptr()) if real_in is not None else None,
                                           C.c_void_p(real_out.data_ptr()) if real_out is not None else None))

    def to_device_real(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    def empty_real(self):
        return self.torch.empty((self.XL, self.ny), dtype=self.torch.float32, device="cuda")

    def set_source(self, src_local):
        if src_local is None:
            self.B.check(self.L.fb_model_set_source(self.model, None))
        else:
            t = self.to_device_real(src_local)
            self.B.check(self.L.fb_model_set_source(self.model, C.c_void_p(t.data_ptr())))
            self.torch.cuda.synchronize()

    def close(self):
        if getattr(self, "model", None):
            self.L.fb_model_destroy(self.model)
            self.L.fb_destroy(self.ctx)
            self.model = None


def _all_to_all(dist, out, inp, world):
    """Equal-split all-to-all of contiguous 1-D tensors; RCCL's collective on GPUs, point-to-point
    pairs elsewhere (gloo has no all_to_all)."""
    if world == 1:
        if out.data_ptr() != inp.data_ptr():
            out.copy_(inp)
        return
    if dist.get_backend() == "nccl":
        dist.all_to_all_single(out, inp)
        return
    rank = dist.get_rank()
    n = inp.numel() // world
    ops = []
    for p in range(world):
        if p == rank:
            out[p * n:(p + 1) * n].copy_(inp[p * n:(p + 1) * n])
        else:
            ops.append(dist.P2POp(dist.isend, inp[p * n:(p + 1) * n], p))
            ops.append(dist.P2POp(dist.irecv, out[p * n:(p + 1) * n], p))
    for r in dist.batch_isend_irecv(ops):
        r.wait()


class SlabModel:
    """RK4 driver on `world` ranks.  API mirrors the single-GPU Model on the rank's local rows."""

    def __init__(self, nx, ny=None, Lx=600000.0, Ly=600000.0, nu=6.5, dt=3.0, rank=0, world=1, backend=None, dist=None):
        ny = ny or nx
        self.nx, self.ny, self.rank, self.world = nx, ny, rank, world
        if dist is None and world > 1:
            import torch.distributed as dist
        self.dist = dist
        self.be = backend if backend is not None else HipBackend(nx, ny, Lx, Ly, nu, dt, rank, world)
        self.XL, self.KS, self.E = self.be.XL, self.be.KS, self.be.E
        self.FL = getattr(self.be, "FL", self.be.E)      # tensor elements per field in the exchange buffers
        self.primed = False

    # -- the two transposes -------------------------------------------------------------------
    def _exchange_w4(self):
        if self.world == 1:
            return
        # [dst][4][XL][KS] -> [src][4][XL][KS]: one collective for the four fields
        _all_to_all(self.dist, self.be.w4_recv, self.be.w4_send, self.world)

    def _exchange_t(self, reverse=False):
        if self.world == 1:
            return
        if reverse:                              # record path: columns (t_recv) -> rows (t_send)
            _all_to_all(self.dist, self.be.t_send, self.be.t_recv, self.world)
        else:
            _all_to_all(self.dist, self.be.t_recv, self.be.t_send, self.world)

    # -- state ------------------------------------------------------------------------------
    def set_vort_local(self, vort_rows):
        """vort_rows: this rank's [XL, ny] rows of the initial vorticity (main.cpp:143-144,256)."""
        be = self.be
        assert tuple(vort_rows.shape) == (self.XL, self.ny)
        d = be.to_device_real(vort_rows)
        be.phase(PH_R2C_ROWS, real_in=d)
        self._exchange_t()
        be.phase(PH_R2C_COLS)
        self.primed = False

    def set_source_local(self, src_rows):
        self.be.set_source(src_rows)

    def vort_local(self):
        """This rank's rows of vort (record path, main.cpp:273-281)."""
        be = self.be
        be.phase(PH_C2R_COLS)
        self._exchange_t(reverse=True)
        out = be.empty_real()
        be.phase(PH_C2R_ROWS, real_out=out)
        return out

    def step(self, n=1):
        be = self.be
        if n <= 0:
            return
        if not self.primed:
            be.phase(PH_PRIME)
            self.primed = True
        for _ in range(n):
            for k in range(4):                   # main.cpp:288-317
                be.phase(PH_COL_BWD)
                self._exchange_w4()
                be.phase(PH_ROW)
                self._exchange_t()
                be.phase(PH_COL_FWD, stage=k)

    def close(self):
        self.be.close()
