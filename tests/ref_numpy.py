"""Independent fp64 numpy restatement of the reference's RK4 step (main.cpp:146-317).

Used ONLY by tests to cross-check the C oracle and to generate tests/golden fixtures.
numpy's rfft2/irfft2 follow the same conventions as FFTW's r2c/c2r (unnormalised forward,
1/N-normalised inverse that drops Im at ky=0 and ky=N/2 -- SURVEY.md note N2).
"""
import numpy as np

TWOPI32 = np.float32(np.arccos(np.float32(-1.0)) * np.float32(2.0))  # fftwfop.hpp:7


def tables(nx, ny, lx, ly):
    """fftwfop.cpp:5-79 restated with numpy scalars of the reference's widths."""
    hx, hy = nx // 2 + 1, ny // 2 + 1
    f32 = np.float32
    gx = np.zeros(nx, dtype=f32)
    for i in range(hx):
        gx[i] = f32(f32(TWOPI32 * f32(i)) / f32(lx))
    for i in range(hx, nx):
        gx[i] = -gx[nx - i]
    gy = np.array([f32(f32(TWOPI32 * f32(j)) / f32(ly)) for j in range(hy)], dtype=f32)
    lap64 = -(gx.astype(np.float64)[:, None] ** 2 + gy.astype(np.float64)[None, :] ** 2)
    lap = lap64.astype(f32)
    for i in range(hx, nx):
        lap[i, :] = lap[nx - i, :]
    lapi = lap.copy()
    lapi[0, 0] = f32(1.0)
    dxw = int(np.ceil(np.float64(f32(nx)) / 3.0))
    dyw = int(np.ceil(np.float64(f32(ny)) / 3.0))
    gws = np.float64(f32(float(dxw) ** 2 + float(dyw) ** 2))
    ii = np.minimum(np.arange(nx), nx - np.arange(nx)).astype(np.float64)
    jj = np.arange(hy).astype(np.float64)
    mask = np.where(ii[:, None] ** 2 + jj[None, :] ** 2 >= gws, 0.0, 1.0).astype(f32)
    return gx, gy, lap, lapi, mask


class Model64:
    """fp64 pseudospectral RK4, same formula order as main.cpp (SURVEY note N4)."""

    def __init__(self, nx, ny, lx=600000.0, ly=600000.0, nu=6.5, dt=3.0):
        self.nx, self.ny = nx, ny
        gx, gy, lap, lapi, mask = tables(nx, ny, lx, ly)
        self.ikx = 1j * gx.astype(np.float64)[:, None]
        self.iky = 1j * gy.astype(np.float64)[None, :]
        self.lap = lap.astype(np.float64)
        self.lapi = lapi.astype(np.float64)
        self.mask = mask.astype(np.float64)
        self.nu = float(np.float32(nu))
        self.dt = float(np.float32(dt))
        self.src = np.zeros((nx, ny))
        self.vc = None

    def set_vort(self, vort):
        self.vc = np.fft.rfft2(vort.astype(np.float64))

    def _c2r(self, c):
        return np.fft.irfft2(c, s=(self.nx, self.ny))  # includes the /GRIDS of main.cpp:37-41

    def tendency(self, vc):
        lv = vc * self.lap
        dzdx = self._c2r(self.ikx * vc)
        dzdy = self._c2r(self.iky * vc)
        psi = vc / self.lapi
        u = -self._c2r(self.iky * psi)
        v = self._c2r(self.ikx * psi)
        t = -u * dzdx - v * dzdy + self.src
        return (np.fft.rfft2(t) + lv * self.nu) * self.mask

    def step(self, n=1):
        dt = self.dt
        for _ in range(n):
            v0 = self.vc
            k1 = self.tendency(v0)
            k2 = self.tendency(v0 + k1 * (dt / 2))
            k3 = self.tendency(v0 + k2 * (dt / 2))
            k4 = self.tendency(v0 + k3 * dt)
            self.vc = v0 + (k1 + 2 * k2 + 2 * k3 + k4) * dt / 6

    def vort(self):
        return self._c2r(self.vc)

    def diag(self):
        psi_c = self.vc / self.lapi
        return self._c2r(psi_c), -self._c2r(self.iky * psi_c), self._c2r(self.ikx * psi_c)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def dft2_r2c_definition(f):
    """FFTW's r2c convention by its DEFINITION, no FFT library involved: Y[i, j] = sum_{x, y} f[x, y] exp(-2 pi i (i x / nx + j y / ny)),
    j = 0 .. ny/2, as two dense matrix products in float64 / complex128 (fftw3 manual, "What FFTW Really Computes")."""
    nx, ny = f.shape
    wx = np.exp(-2j * np.pi * np.outer(np.arange(nx), np.arange(nx)) / nx)
    wy = np.exp(-2j * np.pi * np.outer(np.arange(ny), np.arange(ny // 2 + 1)) / ny)
    return wx @ f.astype(np.float64) @ wy


def dft2_c2r_definition(s, ny):
    """FFTW's c2r_2d on a half spectrum s[nx, ny/2+1] that need NOT be Hermitian (SURVEY note N2), by definition: the complex inverse
    DFT (unnormalised, sign +) along x for every ky column, then per x row the 1-D c2r along y -- the real signal whose half spectrum
    is the row, the imaginary parts at j = 0 and j = ny/2 being ignored: out[y] = Re z0 + (-1)^y Re z_{ny/2} + 2 sum_{0<j<ny/2} Re(z_j e^{+2 pi i j y/ny})."""
    nx = s.shape[0]
    wx = np.exp(+2j * np.pi * np.outer(np.arange(nx), np.arange(nx)) / nx)
    z = wx @ s.astype(np.complex128)                                              # [x][ky]
    j = np.arange(1, ny // 2)
    e = np.exp(+2j * np.pi * np.outer(j, np.arange(ny)) / ny)                       # [j][y]
    out = 2.0 * (z[:, 1:ny // 2] @ e).real
    out += z[:, :1].real
    out += z[:, ny // 2:ny // 2 + 1].real * ((-1.0) ** np.arange(ny))[None, :]
    return out
