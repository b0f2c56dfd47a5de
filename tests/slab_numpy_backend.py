"""CPU test double of the slab compute backend (fp64 numpy), for exercising the exchange logic of
xlab-fftbarotropic_amd/slab.py under gloo.  TEST INFRASTRUCTURE: not part of the product.
Buffer layouts are the contract of include/fftbaro.h ("Slab decomposition")."""
import numpy as np
import torch

import ref_numpy as R

PH_PRIME, PH_COL_BWD, PH_ROW, PH_COL_FWD, PH_R2C_ROWS, PH_R2C_COLS, PH_C2R_COLS, PH_C2R_ROWS = range(8)


class NumpyBackend:
    def __init__(self, nx, ny, Lx, Ly, nu, dt, rank, world):
        from importlib import import_module
        slab = import_module("xlab-fftbarotropic_amd.slab")
        self.nx, self.ny, self.hy, self.rank, self.world = nx, ny, ny // 2 + 1, rank, world
        self.XL, self.KS = slab.slab_geometry(nx, ny, world)
        self.E = nx * self.KS
        self.ky0 = rank * self.KS
        gx, gy, lap, lapi, mask = R.tables(nx, ny, Lx, Ly)
        ptot = self.KS * world
        pad = lambda a: np.concatenate([a, np.zeros((nx, ptot - self.hy))], axis=1)[:, self.ky0:self.ky0 + self.KS]
        self.ikx = 1j * gx.astype(np.float64)[:, None]
        gyp = np.concatenate([gy.astype(np.float64), np.zeros(ptot - self.hy)])
        self.iky = 1j * gyp[None, self.ky0:self.ky0 + self.KS]
        self.lap = pad(lap.astype(np.float64))
        lapi_p = np.concatenate([lapi.astype(np.float64), np.ones((nx, ptot - self.hy))], axis=1)
        self.lapi = lapi_p[:, self.ky0:self.ky0 + self.KS]
        self.mask = pad(mask.astype(np.float64))
        self.nu, self.dt = float(np.float32(nu)), float(np.float32(dt))
        z = lambda n: torch.zeros(n, dtype=torch.complex128)
        self.w4_send, self.w4_recv, self.t_send, self.t_recv = z(4 * self.E), z(4 * self.E), z(self.E), z(self.E)
        if world == 1:
            self.w4_recv = self.w4_send
            self.t_recv = self.t_send
        self.Z = np.zeros((nx, self.KS), dtype=np.complex128)
        self.Z0 = self.Zc = self.acc = None
        self.src = np.zeros((self.XL, ny))

    # helpers ------------------------------------------------------------------------------
    def _derive(self, z):
        psi = z / self.lapi
        d = np.stack([self.ikx * z, self.iky * z, self.iky * psi, self.ikx * psi])     # main.cpp:151,165,198,212
        return d

    def _rows_from_w4(self, f):
        blk = self.w4_recv.numpy().reshape(self.world, 4, self.XL, self.KS)[:, f]       # [src][XL][KS]
        return np.concatenate(list(blk), axis=1)[:, :self.hy]                            # [XL][hy]

    def _rows_from_t_send(self):
        blk = self.t_send.numpy().reshape(self.world, self.XL, self.KS)
        return np.concatenate(list(blk), axis=1)[:, :self.hy]

    def _rows_to_t(self, rows_spec):
        ptot = self.KS * self.world
        full = np.concatenate([rows_spec, np.zeros((self.XL, ptot - self.hy), dtype=np.complex128)], axis=1)
        blk = np.stack([full[:, d * self.KS:(d + 1) * self.KS] for d in range(self.world)])
        self.t_send.view(-1)[:] = torch.from_numpy(np.ascontiguousarray(blk).reshape(-1))

    # phases -------------------------------------------------------------------------------
    def phase(self, ph, stage=0, real_in=None, real_out=None):
        nx, ny = self.nx, self.ny
        if ph == PH_PRIME:
            self.pending = self._derive(self.Z)
            self._stash_block()
        elif ph == PH_COL_BWD:
            pass                                         # the test double does the whole x pass in _stash_block
        elif ph == PH_ROW:
            c2r = lambda f: np.fft.irfft(self._rows_from_w4(f), n=ny, axis=1) * ny / (nx * ny)
            dzdx, dzdy, u, v = c2r(0), c2r(1), -c2r(2), c2r(3)
            t = -u * dzdx - v * dzdy + self.src                                          # main.cpp:225-227
            self._rows_to_t(np.fft.rfft(t, axis=1))
        elif ph == PH_COL_FWD:
            That = np.fft.fft(self.t_recv.numpy().reshape(nx, self.KS), axis=0)
            if stage == 0:
                self.Z0, self.Zc = self.Z, self.Z
            k = (That + self.Zc * self.lap * self.nu) * self.mask                        # main.cpp:148,240-243,296
            dt = self.dt
            if stage == 0:
                self.acc = k; self.Zc = self.Z0 + k * (dt / 2)
            elif stage == 1:
                self.acc = self.acc + 2 * k; self.Zc = self.Z0 + k * (dt / 2)
            elif stage == 2:
                self.acc = self.acc + 2 * k; self.Zc = self.Z0 + k * dt
            else:
                self.Z = self.Z0 + (self.acc + k) * dt / 6; self.Zc = self.Z
            self.pending = self._derive(self.Zc)
            self._stash_block()
        elif ph == PH_R2C_ROWS:
            self._rows_to_t(np.fft.rfft(real_in.numpy().astype(np.float64), axis=1))
        elif ph == PH_R2C_COLS:
            self.Z = np.fft.fft(self.t_recv.numpy().reshape(nx, self.KS), axis=0)
        elif ph == PH_C2R_COLS:
            cols = np.fft.ifft(self.Z, axis=0) * nx                                      # [x][KS] == [dst][XL][KS]
            self.t_recv.view(-1)[:] = torch.from_numpy(np.ascontiguousarray(cols).reshape(-1))
        elif ph == PH_C2R_ROWS:
            real_out.copy_(torch.from_numpy(np.fft.irfft(self._rows_from_t_send(), n=ny, axis=1) * ny / (nx * ny)))
        else:
            raise ValueError(ph)

    def _stash_block(self):
        cols = np.fft.ifft(self.pending, axis=1) * self.nx                               # [4][x][KS]
        blocked = cols.reshape(4, self.world, self.XL, self.KS).transpose(1, 0, 2, 3)    # [dst][4][XL][KS]
        self.w4_send.view(-1)[:] = torch.from_numpy(np.ascontiguousarray(blocked).reshape(-1))

    def to_device_real(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))

    def empty_real(self):
        return torch.empty((self.XL, self.ny), dtype=torch.float64)

    def set_source(self, src_local):
        self.src = np.zeros((self.XL, self.ny)) if src_local is None else np.asarray(src_local, dtype=np.float64)

    def close(self):
        pass
