"""CPU test double of the slab compute backend (fp64 numpy), for exercising the exchange logic of
xlab-fftbarotropic_amd/slab.py under gloo.  TEST INFRASTRUCTURE: not part of the product.

Buffer layouts are the engine's (include/fftbaro.h "Multi-GPU", csrc/fftbaro.hip GroupBufs), per column group g
(0 .. nact-1 = this rank's slabs of the ACTIVE ky columns -- two where a stage is pipelined by column groups --, the last one
= its slab of the FROZEN ones):
    w4_send[g] [dst][4][XL][ncols_g]    w4_recv[g] [src][4][XL][ncols_g]
    t_send[g]  [dst][XL][ncols_g]       t_recv[g]  [nx][ncols_g]
The double is deliberately lazy the way the engine is: a field only reaches w4_send when col_bwd() is asked for it and a
tendency row only reaches t_send when row() is asked for it, so a schedule that forgets a field group or a row chunk
produces wrong numbers instead of passing by accident."""
import numpy as np
import torch

import ref_numpy as R


class NumpyBackend:
    def __init__(self, nx, ny, Lx, Ly, nu, dt, rank, world):
        from importlib import import_module
        slab = import_module("xlab-fftbarotropic_amd.slab")
        self.nx, self.ny, self.hy, self.rank, self.world = nx, ny, ny // 2 + 1, rank, world
        self.XL, self.KA, self.KF = slab.slab_geometry(nx, ny, world)
        act = slab.slab_col_groups(nx, ny, world)
        self.nact = len(act)
        self.ncols = act + ([self.KF] if self.KF else [])
        self.ngroups = len(self.ncols)
        self.katot = world * self.KA
        self.ktot = self.katot + world * self.KF                        # >= hy: columns beyond hy are padding
        # a rank's active slab [rank*KA, (rank+1)*KA) is cut locally into the active groups; frozen slabs follow the active columns
        self.off = [sum(act[:g]) for g in range(self.nact)] + ([0] if self.KF else [])
        self.slab = [self.KA] * self.nact + ([self.KF] if self.KF else [])             # slab width of the group's segment
        self.seg0 = [0] * self.nact + ([self.katot] if self.KF else [])                # first global column of the segment
        self.ky = [self.seg0[g] + rank * self.slab[g] + self.off[g] + np.arange(n) for g, n in enumerate(self.ncols)]
        gx, gy, lap, lapi, mask = R.tables(nx, ny, Lx, Ly)
        padc = lambda a, fill: np.concatenate([a.astype(np.float64), np.full((nx, self.ktot - self.hy), fill)], axis=1)
        self.ikx = 1j * gx.astype(np.float64)[:, None]
        gyp = np.concatenate([gy.astype(np.float64), np.zeros(self.ktot - self.hy)])
        lap_p, lapi_p, mask_p = padc(lap, 0.0), padc(lapi, 1.0), padc(mask, 0.0)
        self.iky = [1j * gyp[None, k] for k in self.ky]
        self.lap = [lap_p[:, k] for k in self.ky]
        self.lapi = [lapi_p[:, k] for k in self.ky]
        self.mask = [mask_p[:, k] for k in self.ky]
        assert not self.KF or not self.mask[self.nact].any(), "the frozen slab must hold masked modes only"
        self.nu, self.dt = float(np.float32(nu)), float(np.float32(dt))
        z = lambda n: torch.zeros(n, dtype=torch.complex128)
        self.w4_send = [z(4 * nx * n) for n in self.ncols]
        self.t_send = [z(nx * n) for n in self.ncols]
        if world == 1:
            self.w4_recv, self.t_recv = self.w4_send, self.t_send
        else:
            self.w4_recv = [z(4 * nx * n) for n in self.ncols]
            self.t_recv = [z(nx * n) for n in self.ncols]
        self.Z = [np.zeros((nx, n), dtype=np.complex128) for n in self.ncols]       # vort_c, [kx][local ky]
        self.Z0, self.Zc, self.acc = [None] * self.nact, [None] * self.nact, [None] * self.nact
        self.pending = [None] * self.nact                                             # active groups: x-transformed derivative fields not yet in w4_send
        self.src = np.zeros((self.XL, ny))

    # helpers ------------------------------------------------------------------------------
    def _derive(self, z, g):
        psi = z / self.lapi[g]
        return np.stack([self.ikx * z, self.iky[g] * z, self.iky[g] * psi, self.ikx * psi])     # main.cpp:151,165,198,212

    def _blocked(self, fields_x, g):
        """[4][x][ncols] -> flat [dst][4][XL][ncols]"""
        n = self.ncols[g]
        return np.ascontiguousarray(fields_x.reshape(4, self.world, self.XL, n).transpose(1, 0, 2, 3)).reshape(-1)

    def _rows(self, bufs, nfields, f):
        """field f of the row-side buffers of every group -> [XL][hy]"""
        full = np.zeros((self.XL, self.ktot), dtype=np.complex128)
        for g, n in enumerate(self.ncols):
            if n:
                blk = bufs[g].numpy().reshape(self.world, nfields, self.XL, n)[:, f]                   # [src][XL][ncols]
                for s_ in range(self.world):
                    c0 = self.seg0[g] + s_ * self.slab[g] + self.off[g]
                    full[:, c0:c0 + n] = blk[s_]
        return full[:, :self.hy]

    def _rows_to_t(self, rows_spec, x0, frozen):
        """half-spectrum rows [nrows][hy] of local rows x0.. -> t_send, blocked by destination rank"""
        nrows = rows_spec.shape[0]
        full = np.concatenate([rows_spec, np.zeros((nrows, self.ktot - self.hy), dtype=np.complex128)], axis=1)
        for g in range(self.ngroups if frozen else self.nact):
            n = self.ncols[g]
            if not n:
                continue
            view = self.t_send[g].numpy().reshape(self.world, self.XL, n)
            for d in range(self.world):
                c0 = self.seg0[g] + d * self.slab[g] + self.off[g]
                view[d, x0:x0 + nrows] = full[:, c0:c0 + n]

    # phases (names follow csrc/fb_slab_driver.h) -------------------------------------------
    def prime(self):
        for g in range(self.ngroups):
            if not self.ncols[g]:
                continue
            cols = np.fft.ifft(self._derive(self.Z[g], g), axis=1) * self.nx                       # [4][x][ncols]
            if g < self.nact:
                self.pending[g] = cols
                self.w4_send[g].zero_()
            else:
                self.w4_send[g].numpy()[:] = self._blocked(cols, g)                                # frozen columns: final, sent once

    def col_bwd(self, f0, f1, g=0):
        n = self.ncols[g]
        view = self.w4_send[g].numpy().reshape(self.world, 4, self.XL, n)
        view[:, f0:f1] = self.pending[g][f0:f1].reshape(f1 - f0, self.world, self.XL, n).transpose(1, 0, 2, 3)

    def row(self, x0, nrows):
        nx, ny = self.nx, self.ny
        c2r = lambda f: np.fft.irfft(self._rows(self.w4_recv, 4, f)[x0:x0 + nrows], n=ny, axis=1) * ny / (nx * ny)
        dzdx, dzdy, u, v = c2r(0), c2r(1), -c2r(2), c2r(3)
        t = -u * dzdx - v * dzdy + self.src[x0:x0 + nrows]                                        # main.cpp:225-227
        self._rows_to_t(np.fft.rfft(t, axis=1), x0, frozen=False)

    def col_fwd(self, stage, g=0):
        nx = self.nx
        That = np.fft.fft(self.t_recv[g].numpy().reshape(nx, self.ncols[g]), axis=0)
        if stage == 0:
            self.Z0[g], self.Zc[g] = self.Z[g], self.Z[g]
        k = (That + self.Zc[g] * self.lap[g] * self.nu) * self.mask[g]                          # main.cpp:148,240-243,296
        dt = self.dt
        if stage == 0:
            self.acc[g] = k; self.Zc[g] = self.Z0[g] + k * (dt / 2)
        elif stage == 1:
            self.acc[g] = self.acc[g] + 2 * k; self.Zc[g] = self.Z0[g] + k * (dt / 2)
        elif stage == 2:
            self.acc[g] = self.acc[g] + 2 * k; self.Zc[g] = self.Z0[g] + k * dt
        else:
            self.Z[g] = self.Z0[g] + (self.acc[g] + k) * dt / 6; self.Zc[g] = self.Z[g]
        self.pending[g] = np.fft.ifft(self._derive(self.Zc[g], g), axis=1) * nx
        self.t_recv[g].zero_()                                                                   # consumed: a stale re-read would show
        if self.world > 1:
            self.w4_send[g].zero_()

    def r2c_rows(self, real_rows):
        self._rows_to_t(np.fft.rfft(np.asarray(real_rows, dtype=np.float64), axis=1), 0, frozen=True)

    def r2c_cols(self):
        for g in range(self.ngroups):
            if self.ncols[g]:
                self.Z[g] = np.fft.fft(self.t_recv[g].numpy().reshape(self.nx, self.ncols[g]), axis=0)

    def c2r_cols(self):
        for g in range(self.ngroups):
            if self.ncols[g]:
                cols = np.fft.ifft(self.Z[g], axis=0) * self.nx                                   # [x][ncols] == [dst][XL][ncols]
                self.t_recv[g].numpy()[:] = np.ascontiguousarray(cols).reshape(-1)

    def c2r_rows(self):
        rows = self._rows(self.t_send, 1, 0)
        return torch.from_numpy(np.fft.irfft(rows, n=self.ny, axis=1) * self.ny / (self.nx * self.ny))

    def set_source(self, src_local):
        self.src = np.zeros((self.XL, self.ny)) if src_local is None else np.asarray(src_local, dtype=np.float64)

    def close(self):
        pass
