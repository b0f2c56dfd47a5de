"""GPU test of the slab-decomposed kernels: all `world` ranks' local passes run on the ONE GPU of
the box, the all-to-all transposes are emulated by device copies with the same block semantics
RCCL's all_to_all_single has.  Checks the SLAB addressing of the row kernel, the ky0 offsets of
the column kernels and the phase order against the fused single-GPU path (bit for bit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _emulated_all_to_all(recv_list, send_list, world, n_per_block, offset=0):
    for d in range(world):
        for s in range(world):
            recv_list[d][offset + s * n_per_block: offset + (s + 1) * n_per_block].copy_(
                send_list[s][offset + d * n_per_block: offset + (d + 1) * n_per_block])


class _Group:
    def __init__(self, X, slab, n, world, dt, src=None):
        self.world, self.slab = world, slab
        self.be = [slab.HipBackend(n, n, 6e5, 6e5, 6.5, dt, r, world) for r in range(world)]
        self.E = self.be[0].FL              # tensor elements per field (float32 view)
        self.blk = self.E // world

    def xw4(self):
        if self.world == 1:
            return
        _emulated_all_to_all([b.w4_recv for b in self.be], [b.w4_send for b in self.be], self.world, 4 * self.blk)

    def xt(self, reverse=False):
        if self.world == 1:
            return
        if reverse:
            _emulated_all_to_all([b.t_send for b in self.be], [b.t_recv for b in self.be], self.world, self.blk)
        else:
            _emulated_all_to_all([b.t_recv for b in self.be], [b.t_send for b in self.be], self.world, self.blk)

    def ph(self, ph, **kw):
        for b in self.be:
            b.phase(ph, **kw)


@pytest.mark.parametrize("world,n,steps", [(1, 256, 3), (2, 256, 3), (4, 256, 2), (8, 512, 1), (2, 1024, 2), (4, 768, 2)])
def test_slab_phases_match_fused_path(world, n, steps):
    import torch
    from importlib import import_module
    import xlab_fftbarotropic_amd as X
    slab = import_module("xlab-fftbarotropic_amd.slab")
    S = slab
    dt = 3.0
    v0 = X.make_field("elliptic", n)
    src = X.make_source_kuo2004(n)
    ref = X.Model(n, n, dt=dt)
    ref.set_vort(v0)
    ref.set_source(src)
    ref.step(steps)
    want = ref.vort().cpu().numpy()

    g = _Group(X, slab, n, world, dt)
    for r, b in enumerate(g.be):
        b.set_source(S.local_rows(src, r, world))
        b.phase(S.PH_R2C_ROWS, real_in=b.to_device_real(S.local_rows(v0, r, world)))
    g.xt()
    g.ph(S.PH_R2C_COLS)
    g.ph(S.PH_PRIME)
    for _ in range(steps):
        for k in range(4):
            g.ph(S.PH_COL_BWD)
            g.xw4()
            g.ph(S.PH_ROW)
            g.xt()
            g.ph(S.PH_COL_FWD, stage=k)
    g.ph(S.PH_C2R_COLS)
    g.xt(reverse=True)
    rows = []
    for b in g.be:
        out = b.empty_real()
        b.phase(S.PH_C2R_ROWS, real_out=out)
        rows.append(out.cpu().numpy())
    got = np.concatenate(rows, axis=0)
    torch.cuda.synchronize()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for b in g.be:
        b.close()


def test_slab_model_world1_api():
    """SlabModel with world == 1 (no process group) equals Model."""
    from importlib import import_module
    import xlab_fftbarotropic_amd as X
    slab = import_module("xlab-fftbarotropic_amd.slab")
    n = 256
    v0 = X.make_field("gaussian", n)
    m = slab.SlabModel(n, n, rank=0, world=1)
    m.set_vort_local(v0)
    m.step(4)
    ref = X.Model(n, n)
    ref.set_vort(v0)
    ref.step(4)
    assert np.array_equal(m.vort_local().cpu().numpy().view(np.uint32), ref.vort().cpu().numpy().view(np.uint32))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_on_one_gpu_gloo(tmp_path):
    """The real multi-process flow of bench.py (torch.distributed.run, SlabModel + HipBackend, exchange
    between phases) with 2 ranks sharing the GPU over gloo -- RCCL itself needs two devices."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--grid", "512", "--backend", "gloo", "--cpu-steps", "0"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "slab2"


def test_slab_model_two_ranks_gloo_matches_single(tmp_path):
    """SlabModel on 2 processes (one GPU, gloo) reproduces the single-process field bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from importlib import import_module\n"
        "import xlab_fftbarotropic_amd as X\n"
        "slab = import_module('xlab-fftbarotropic_amd.slab')\n"
        "torch.cuda.set_device(0); dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size(); n = 256\n"
        "v0 = X.make_field('elliptic', n)\n"
        "m = slab.SlabModel(n, n, rank=r, world=w); m.set_vort_local(slab.local_rows(v0, r, w)); m.step(3)\n"
        "np.save(os.path.join(%r, 'rows%%d.npy' %% r), m.vort_local().cpu().numpy())\n"
        "dist.destroy_process_group()\n" % (root, str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    import xlab_fftbarotropic_amd as X
    n = 256
    ref = X.Model(n, n)
    ref.set_vort(X.make_field("elliptic", n))
    ref.step(3)
    want = ref.vort().cpu().numpy()
    got = np.concatenate([np.load(str(tmp_path / ("rows%d.npy" % r))) for r in range(2)], axis=0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
