"""CPU-only: the C-ABI library loads and exports every declared symbol; host-side weight packing matches
the MFMA fragment convention the kernel assumes; checkpoint layout conversions; rank sharding and the
confusion-counter all_gather on gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from insmos_amd import _lib
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from insmos_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "insmos_hip.h")).read()
    declared = set(re.findall(r"\b(insmos_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.insmos_version() >= 100


def _emulate_conv(packed, K, cin, cout, x, nbr):
    """Numpy emulation of k_sparse_conv's contraction straight from the PACKED weights: one MFMA step s of a
    block contracts, for lane group g, input channel c0 + width*g + s with A = packed[k][blk][tile][lane][s]
    (4 floats per lane; Cin = 8 / 4 layers -- one chunk of width 2 / 1 -- store just those 2 / 1)."""
    n16, rem = cin // 16, cin % 16
    blocks = [(c * 16, 4) for c in range(n16)]
    c0 = n16 * 16
    if rem & 8:
        blocks.append((c0, 2)); c0 += 8
    if rem & 4:
        blocks.append((c0, 1))
    ntile = (cout + 15) // 16
    lf = 2 if cin == 8 else 1 if cin == 4 else 4   # floats per lane: the single-chunk layers (Cin = 8 / 4) are stored compactly
    pk = packed[:K * len(blocks) * ntile * 64 * lf].reshape(K, len(blocks), ntile, 64, lf)   # (a row-lane tail may follow)
    n_out = nbr.shape[1]
    out = np.zeros((n_out, ntile * 16), np.float64)
    for k in range(K):
        valid = nbr[k] >= 0
        rows = x[np.where(valid, nbr[k], 0)] * valid[:, None]
        for b, (cb, width) in enumerate(blocks):
            for s in range(width):
                for g in range(4):
                    ch = cb + width * g + s
                    for t in range(ntile):
                        a = pk[k, b, t, g * 16:(g + 1) * 16, s]  # A[i = lane&15] for this lane group
                        out[:, t * 16:(t + 1) * 16] += rows[:, ch:ch + 1] * a[None, :]
    return out[:, :cout]


@pytest.mark.parametrize("cin_real,cout_real,K", [(8, 8, 5), (1, 8, 3), (7, 16, 4), (19, 16, 3), (35, 32, 2), (48, 32, 2),
                                                  (24, 16, 3), (131, 24, 2), (16, 3, 1), (128, 11, 1), (16, 16, 3), (13, 8, 2),
                                                  (8, 16, 1)])
def test_weight_packing_matches_fragment_convention(lib, cin_real, cout_real, K):
    from oracle import ref_ops as R
    rng = np.random.default_rng(cin_real * 7 + cout_real)
    cin = (cin_real + 3) // 4 * 4
    cout = cout_real if cout_real % 4 == 0 else (cout_real + 3) // 4 * 4
    taps = rng.normal(size=(K, cin_real, cout_real)).astype(np.float32)
    n = lib.insmos_packed_weight_floats(K, cin, cout)
    packed = np.full(n, np.nan, np.float32)
    assert lib.insmos_pack_weights_host(taps.ctypes.data, K, cin_real, cout_real, cin, cout, packed.ctypes.data) == 0
    assert not np.isnan(packed).any()
    n_in, n_out = 40, 33
    x = np.zeros((n_in, cin), np.float32)
    x[:, :cin_real] = rng.normal(size=(n_in, cin_real))
    x[:, cin_real:] = 3.0  # padded channels carry junk; their taps must be zero
    nbr = rng.integers(-1, n_in, size=(K, n_out)).astype(np.int32)
    got = _emulate_conv(packed, K, cin, cout, x.astype(np.float64), nbr)
    ref = R.sparse_conv(x[:, :cin_real], nbr, taps)
    np.testing.assert_allclose(got[:, :cout_real], ref, rtol=1e-4, atol=1e-4)
    assert np.all(got[:, cout_real:] == 0)
    # the row-lane tail of the small-channel layers (csrc/spconv_rowlane.hip): [tap][p][co] behind the fragments, p = 4 s + g
    # walking the input channels in the MFMA chain order ci = (cin / 4) g + s; no tail for K = 1 or other widths
    frag = K * ((cin // 16) + (1 if cin % 16 >= 8 else 0) + (1 if cin % 8 >= 4 else 0)) * ((cout + 15) // 16) * 64 * (2 if cin == 8 else 1 if cin == 4 else 4)
    if cin in (8, 16) and cout in (8, 16) and K >= 2:
        assert n == frag + K * cin * cout
        tail = packed[frag:].reshape(K, cin, cout)
        for p in range(cin):
            ci = (cin // 4) * (p & 3) + (p >> 2)
            want = np.zeros((K, cout), np.float32)
            if ci < cin_real:
                want[:, :cout_real] = taps[:, ci, :]
            np.testing.assert_array_equal(tail[:, p, :], want)
    else:
        assert n == frag
    assert lib.insmos_pack_weights_host(taps.ctypes.data, K, cin_real, cout_real, cin + 1, cout, packed.ctypes.data) == -1


def test_checkpoint_spec_and_layout_conversions():
    from insmos_amd import params as P
    cfg = P.default_cfg()
    sd = P.random_state_dict(cfg, 3)
    spec = P.param_spec(cfg)
    assert list(sd) == list(spec)
    for k, (shape, _) in spec.items():
        assert sd[k].shape == tuple(shape) and sd[k].dtype == np.float32
    assert sum(v.size for v in sd.values()) > 4_000_000  # ~4.8 M UNetV2 + BEV + MotionNet parameters
    w = np.arange(2 * 3 * 1 * 2 * 5, dtype=np.float32).reshape(2, 3, 1, 2, 5)  # (Cout,kz,ky,kx,Cin)
    t = P.spconv_weight_to_taps(w)
    assert t.shape == (6, 5, 2)
    assert t[3, 4, 1] == w[1, 1, 0, 1, 4]  # k = (kz*KH + ky)*KW + kx = (1*1+0)*2+1 = 3
    w2 = np.arange(4 * 3 * 3 * 3, dtype=np.float32).reshape(4, 3, 3, 3)
    assert P.conv2d_weight_to_taps(w2)[5, 2, 1] == w2[1, 2, 1, 2]  # tap = ky*3 + kx
    wt = np.arange(3 * 4 * 2 * 2, dtype=np.float32).reshape(3, 4, 2, 2)
    assert P.convT2d_weight_to_taps(wt)[2, 1, 3] == wt[1, 3, 1, 0]
    taps, shift = P.fold_bn(np.ones((1, 2, 3), np.float32), [2, 2, 2], [1, 1, 1], [0.5, 0.5, 0.5], [3, 3, 3], 1.0)
    np.testing.assert_allclose(taps, 1.0)  # 2/sqrt(3+1) = 1
    np.testing.assert_allclose(shift, 0.5)


def test_models_surface_fails_loudly_without_gpu(tmp_path):
    import torch
    from insmos_amd import params as P
    from insmos_amd.models import InsMOSNet, save_checkpoint
    cfg = P.default_cfg()
    path = str(tmp_path / "x.ckpt")
    save_checkpoint(path, cfg, P.random_state_dict(cfg, 0))
    m = InsMOSNet.load_from_checkpoint(path, hparams=cfg)
    assert m.n_mos_classes == 3 and m.ignore_index == [0]
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.cuda()
    with pytest.raises(KeyError):
        InsMOSNet(cfg, state_dict={})


def test_shard_indices():
    from insmos_amd.metrics import shard_indices
    parts = [shard_indices(10, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(10))
    assert parts[1] == [1, 5, 9]


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from insmos_amd.metrics import all_gather_confusion, shard_indices
rank, world = int(sys.argv[1]), 2
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
cm = torch.zeros((3, 3), dtype=torch.int64)
for w in shard_indices(7, rank, world):          # windows owned by this rank
    cm[w % 3, (w * 2) % 3] += w + 1
tot = all_gather_confusion(cm)
exp = torch.zeros((3, 3), dtype=torch.int64)
for w in range(7):
    exp[w % 3, (w * 2) % 3] += w + 1
assert torch.equal(tot, exp), (tot, exp)
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
"""


def test_confusion_all_gather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, ROOT], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


_DDP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from insmos_amd.ddp import BucketedGradReducer
rank, world = int(sys.argv[1]), 2
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
params = {"w%02d" % i: torch.zeros(s, requires_grad=True) for i, s in enumerate([(81, 8, 8), (8,), (27, 16, 32), (3,), (125, 1, 8), (1, 8, 3)])}
grads = {r: {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(100 * r + i)) for i, (k, v) in enumerate(sorted(params.items()))}
         for r in range(world)}
for k, v in params.items():
    if not (k == "w03" and rank == 1):          # a parameter without a gradient on one rank counts as zero
        v.grad = grads[rank][k].clone()
red = BucketedGradReducer(params, bucket_bytes=20000)   # several buckets, one tensor larger than a bucket
n = red.reduce(average=True)
assert n >= 3, n
for k, v in params.items():
    exp = (grads[0][k] + (torch.zeros_like(grads[1][k]) if k == "w03" else grads[1][k])) / 2
    assert torch.allclose(v.grad, exp, atol=1e-7), k
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
"""


def test_bucketed_grad_reducer_gloo_world2(tmp_path):
    """The DDP gradient exchange of the training slice (insmos_amd/ddp.py): bucketed all-reduce == per-tensor mean."""
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER)
    port = str(29900 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, ROOT], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


_DDP_OVERLAP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from insmos_amd.ddp import BucketedGradReducer
rank, world = int(sys.argv[1]), 2
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
shapes = [(40, 8), (8,), (30, 16), (16,), (20, 4), (4,), (9, 3)]          # forward order: layer 0 first
params = {"p%d" % i: torch.nn.Parameter(torch.randn(s)) for i, s in enumerate(shapes)}
red = BucketedGradReducer(params, bucket_bytes=1500, overlap=True)       # several buckets, laid out last layer first
assert len(red.buckets) >= 3 and red.buckets[0][1][0][0] is params["p6"]
def loss_fn(r, step):
    x = torch.full((5, 40), 0.1 * (r + 1) + step)
    h = torch.tanh(x @ params["p0"] + params["p1"])
    h = torch.tanh(h @ params["p2"][:8] + params["p3"])
    h = h @ params["p4"][:16] + params["p5"]
    out = h.sum()
    if r == 0:                       # rank 1 never touches p6: its gradient counts as zero there
        out = out + (params["p6"] ** 2).sum()
    return out
for step in range(2):                # the reducer is reusable step after step
    for p in params.values():
        p.grad = None
    loss_fn(rank, step).backward()
    if rank == 0:
        assert red._launched >= 1    # a bucket went out DURING backward
    n = red.reduce(average=True)
    assert n == len(red.buckets)
    # reference: both ranks' gradients computed locally, averaged
    exp = {k: torch.zeros_like(v) for k, v in params.items()}
    saved = {k: v.grad.clone() for k, v in params.items()}
    for r in range(world):           # (autograd.grad: no accumulation into .grad, so the reducer's hooks stay out of it)
        gs = torch.autograd.grad(loss_fn(r, step), list(params.values()), allow_unused=True)
        for k, g in zip(params, gs):
            if g is not None:
                exp[k] += g / world
    for k in params:
        assert torch.allclose(saved[k], exp[k], atol=1e-6), (step, k)
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
"""


def test_overlapped_grad_reducer_gloo_world2(tmp_path):
    """overlap=True: buckets go out during backward, in bucket order on every rank, a rank's missing gradient counts as zero."""
    script = tmp_path / "ddp_overlap_worker.py"
    script.write_text(_DDP_OVERLAP_WORKER)
    port = str(29100 + os.getpid() % 150)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, ROOT], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_unet_trainer_layouts_round_trip_and_deconv_table():
    """Host side of insmos_amd/train_unet.py (no GPU): checkpoint layout -> tap layout -> export is the identity, and the
    4-tap table that stands for ConvTranspose2d(2, 2) reproduces torch's conv_transpose2d."""
    import types
    import torch
    import torch.nn.functional as F
    from insmos_amd import params as P
    from insmos_amd.train_unet import UNetV2Trainer
    cfg = P.default_cfg()
    sd = P.random_state_dict(cfg, 3)
    H, W = 3, 5
    tr = UNetV2Trainer(cfg, sd, device="cpu", engine=types.SimpleNamespace(bevH=H, bevW=W))
    exp = tr.export_state_dict()
    unet_keys = [k for k in P.param_spec(cfg) if k.startswith(P.UNET_PREFIX)]
    assert sorted(exp) == sorted(unet_keys)
    for k in unet_keys:
        np.testing.assert_array_equal(exp[k], np.asarray(sd[k], np.float32), err_msg=k)
    assert all(not v.requires_grad for v in tr.buffers.values()) and all(v.requires_grad for v in tr.params.values())
    nbr, nbr_t = tr._deconv_nbr()
    assert nbr.shape == (4, 4 * H * W) and nbr_t.shape == (4, H * W)
    assert int((nbr >= 0).sum()) == 4 * H * W and int((nbr_t >= 0).sum()) == 4 * H * W
    for k in range(4):
        o = torch.nonzero(nbr[k] >= 0).flatten()
        assert torch.equal(nbr_t[k][nbr[k][o].long()].long(), o)   # transposed table: nbr_t[k][i] = o <=> nbr[k][o] = i
    rng = np.random.default_rng(0)
    ci, co = 6, 4
    wt = torch.from_numpy(rng.normal(size=(ci, co, 2, 2)).astype(np.float32))
    x = torch.from_numpy(rng.normal(size=(H * W, ci)).astype(np.float32))
    taps = torch.from_numpy(P.convT2d_weight_to_taps(wt.numpy()))
    y = torch.zeros((4 * H * W, co))
    for k in range(4):
        o = torch.nonzero(nbr[k] >= 0).flatten()
        y[o] += x[nbr[k][o].long()] @ taps[k]
    ref = F.conv_transpose2d(x.T.reshape(1, ci, H, W), wt, stride=2)[0].permute(1, 2, 0).reshape(4 * H * W, co)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=1e-5)


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors in insmos_amd/_lib.py against include/insmos_hip.h as a C compiler lays it out: size of every
    struct and offset of every field (a silent mismatch would corrupt the native runner's configuration / results)."""
    import ctypes
    import subprocess
    from insmos_amd import _lib
    pairs = {"InsmosConvW": _lib.ConvW, "InsmosNetCfg": _lib.NetCfg, "InsmosForwardOut": _lib.ForwardOut, "InsmosRankJob": _lib.RankJob}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "insmos_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
        assert len([k for k in got if k[0] == cname]) == len(cls._fields_) + 1


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/insmos_hip.h against insmos_amd._lib.SIGNATURES: return type, argument count and the ctypes
    class of every argument (a pointer passed as int, or an int64 as int, fails silently at run time)."""
    import ctypes
    import re
    from insmos_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "insmos_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"typedef struct.*?\}\s*\w+;", "", hdr, flags=re.S)

    def ctype(decl):
        d = decl.strip()
        if "*" in d:
            return ctypes.c_void_p
        base = " ".join(t for t in d.split()[:-1] if t != "const") if len(d.split()) > 1 else d
        return {"int": ctypes.c_int, "int32_t": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
                "double": ctypes.c_double, "size_t": ctypes.c_size_t, "unsigned": ctypes.c_uint, "uint32_t": ctypes.c_uint,
                "unsigned int": ctypes.c_uint}[base]

    protos = re.findall(r"\b(int|size_t|const char\*|void)\s+(insmos_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 60
    seen = set()
    for ret, name, args in protos:
        seen.add(name)
        if name not in _lib.SIGNATURES:
            continue   # (name coverage is test_header_symbols_all_exported's job)
        restype, argtypes = _lib.SIGNATURES[name]
        want_ret = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p, "void": None}[ret]
        assert restype == want_ret, (name, restype, ret)
        decls = [a for a in (x.strip() for x in args.replace("\n", " ").split(",")) if a and a != "void"]
        assert len(decls) == len(argtypes), (name, len(decls), len(argtypes))
        for i, (dcl, at) in enumerate(zip(decls, argtypes)):
            assert ctype(dcl) == at, (name, i, dcl, at)
    assert seen >= set(_lib.SIGNATURES), set(_lib.SIGNATURES) - seen


_BENCH_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
import bench
rank, world = int(sys.argv[1]), 2
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
cores = bench.pin_host_threads(rank, world)
assert cores >= 1 and torch.get_num_threads() >= 1
W, steps = 3, 4
class StubMetrics:
    def compute_confusion_matrix(self, lg, gt, out=None):
        pred = lg.argmax(1)
        for p, g in zip(pred.tolist(), gt.tolist()):
            out[p, g] += 1
        return out
calls = []
def forward(batch, mode):                        # rank 1 is the slow rank: the job's time is ITS time
    calls.append(len(batch))
    time.sleep(0.02 if rank == 0 else 0.06)
    logits = [torch.eye(3)[(torch.arange(5) + i + rank) % 3] for i in range(len(batch))]
    return None, None, logits
batch = [{"past_point_clouds": None}] * W
gts = [(torch.arange(5) + i) % 3 for i in range(W)]
dt, value, cm_all = bench.timed_steps(forward, batch, gts, StubMetrics(), steps, 2, world, "cpu", lambda: None)
assert len(calls) == steps + 2 and all(c == W for c in calls)
assert dt >= steps * 0.06 * 0.95, dt                      # MAX over ranks, on both ranks
assert abs(value - world * steps * W / dt) < 1e-9          # whole-job aggregate
exp = torch.zeros((3, 3), dtype=torch.int64)
for r in range(world):
    for i in range(W):
        for k in range(5):
            exp[(k + i + r) % 3, (k + i) % 3] += steps
assert torch.equal(cm_all, exp), (cm_all, exp)             # counters of BOTH ranks, timed steps only
lo, hi = bench.timed_steps.last_rank_ms                    # per-rank spread of a rank's OWN ms per step (rank_ms_min / rank_ms_max):
assert 18.0 <= lo <= 45.0 and 58.0 <= hi <= 1000.0 * dt / steps + 1.0, (lo, hi, dt)   # the straggler (rank 1) is visible on both ranks
dist.barrier(); dist.destroy_process_group()
print("OK", rank, round(dt, 3))
"""


def test_bench_timed_region_shard_and_aggregate_gloo_world2(tmp_path):
    """bench.py's contract pieces without a GPU: two ranks over gloo with a stub model -- warm-up untimed, exactly K timed
    steps, time = MAX over ranks, value = whole-job windows / that time, confusion counters all-gathered inside the region."""
    script = tmp_path / "bench_worker.py"
    script.write_text(_BENCH_WORKER)
    port = str(29300 + os.getpid() % 200)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, ROOT], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_checkpoint_layout_check_and_first_contact_warning(tmp_path):
    """load_from_checkpoint's layout gate (no GPU needed: it runs before the engine is built) and tools/ckpt_probe.py: a
    checkpoint written by this package passes silently; the same tensors without the marker -- what a real Lightning
    checkpoint looks like -- warn loudly that the ME / spconv layout conversions are unverified; an spconv 1.x-shaped weight
    (kz, ky, kx, Cin, Cout) is refused with the probe's diagnosis instead of being loaded with permuted taps."""
    import warnings
    import torch
    from insmos_amd import models, params as P
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ckpt_probe
    cfg = P.default_cfg()
    sd = {k: torch.as_tensor(v) for k, v in P.random_state_dict(cfg, 0).items()}
    ok = {"hyper_parameters": cfg, "state_dict": sd, "insmos_amd_synthetic": True}
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        models._check_checkpoint_layouts(ok, cfg, "synthetic.ckpt")
    with pytest.warns(RuntimeWarning, match="first-contact"):
        models._check_checkpoint_layouts({"hyper_parameters": cfg, "state_dict": sd}, cfg, "real.ckpt")
    n_ok, problems = ckpt_probe.probe(ok, out=lambda *_: None)
    assert not problems and n_ok == len(P.param_spec(cfg))
    key = P.UNET_PREFIX + "conv2.1.0.weight"
    old = dict(sd)
    old[key] = sd[key].permute(1, 2, 3, 4, 0).contiguous()       # (Cout, kz, ky, kx, Cin) -> (kz, ky, kx, Cin, Cout)
    with pytest.raises(ValueError, match="ckpt_probe"):
        models._check_checkpoint_layouts({"hyper_parameters": cfg, "state_dict": old, "insmos_amd_synthetic": True}, cfg, "old.ckpt")
    lines = []
    _, problems = ckpt_probe.probe({"hyper_parameters": cfg, "state_dict": old}, out=lines.append)
    assert len(problems) == 1 and "spconv 1.x" in problems[0][1], problems
    del old[P.ME_PREFIX + "block1.0.conv1.kernel"]
    _, problems = ckpt_probe.probe({"hyper_parameters": cfg, "state_dict": old}, out=lines.append)
    assert any(p[1] == "MISSING" for p in problems)


def test_bn_plan_chunk_tables_cover_every_row_once_per_segment():
    """BnPlan (insmos_amd/autograd.py): the chunk table the segmented BatchNorm kernels walk -- chunks of one segment each, sorted by
    segment, covering every row exactly once, segment row counts right; one table per chunk length, built on demand."""
    import torch
    from insmos_amd.autograd import BnPlan
    seg = torch.tensor([0] * 2500 + [1] * 700 + [0] * 1030 + [2] * 5)
    plan = BnPlan.from_segment_ids(seg, 4)                     # segment 3 is empty
    assert plan.S == 4 and plan.n_rows == len(seg) and plan.seg_rows.tolist() == [3530, 700, 5, 0]
    for c in (8, 128):
        t = plan.table(c)
        ch = t.chunks.numpy()
        assert t.n_chunks == len(ch) and (ch[:, 1] > ch[:, 0]).all() and (ch[:, 1] - ch[:, 0] <= BnPlan.chunk_rows(c)).all()
        assert (np.diff(ch[:, 2]) >= 0).all()                  # sorted by segment
        cover = np.zeros(len(seg), np.int32)
        for r0, r1, sg, _ in ch:
            assert (seg[r0:r1] == sg).all()                    # a chunk never straddles segments
            cover[r0:r1] += 1
        assert (cover == 1).all()
        first = t.seg_first.tolist()
        assert first[0] == 0 and first[-1] == t.n_chunks and all(a <= b for a, b in zip(first, first[1:]))
        assert first[3] == first[4]                            # the empty segment has no chunks
    assert plan.table(8) is plan.table(16) and BnPlan.chunk_rows(128) == 512 and BnPlan.chunk_rows(8) == 4096   # same length -> same table
    whole = BnPlan.whole(3000, "cpu")
    assert whole.S == 1 and whole.table(64).n_chunks == 3 and whole.table(8).n_chunks == 1 and whole.seg_rows.tolist() == [3000]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus N` with no launcher around it starts N ranks (round-3 review: the flag was parsed and ignored, an
    8-GPU call would have measured one GPU); inside a launcher a --gpus that disagrees with WORLD_SIZE is refused.  The
    rendezvous-only mode runs the real spawn + process group over gloo without touching a GPU."""
    import json
    bench_py = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench_py, "--gpus", "2", "--backend", "gloo", "--rendezvous-only"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:]
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["backend"] == "gloo", out[-2000:]   # ONE line, from rank 0
    r = subprocess.run([sys.executable, bench_py, "--gpus", "2", "--rendezvous-only"], env=dict(env, WORLD_SIZE="4", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stdout.decode()
    import bench
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "5"], port=1234)
    assert cmd[1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                        "--master-port", "1234"] and cmd[-4:] == ["--gpus", "8", "--steps", "5"]


def test_precision_modes_are_refused_unless_known():
    """The reduced-precision switches are opt-in and validated on the host: the library refuses unknown modes (and stays in
    exact fp32), the training switch accepts only 0 / 1, and the d/dW kernel selector only its three kernels."""
    from insmos_amd import _lib, autograd
    lib = _lib.load()
    assert lib.insmos_conv_precision(0) == 0
    for bad in (2, 4, -1, 7):
        assert lib.insmos_conv_precision(bad) != 0
    assert lib.insmos_register_split_weights(None, None) != 0
    for bad in (2, 3, -1):
        with pytest.raises(ValueError):
            autograd.set_train_conv_precision(bad)
    autograd.set_train_conv_precision(0)
    assert lib.insmos_debug_dw_kernel(3) != 0 and lib.insmos_debug_dw_kernel(-1) != 0 and lib.insmos_debug_dw_kernel(2) == 0
    # the workspace plan follows the kernel: smaller row chunks (more partials) for small layers under the row-compacting kernel
    small = lib.insmos_sparse_conv_backward_weight_ws_floats(5000, 27, 128, 128)
    assert lib.insmos_debug_dw_kernel(0) == 0
    assert lib.insmos_sparse_conv_backward_weight_ws_floats(5000, 27, 128, 128) == 2 * 27 * 128 * 128 <= small
    assert lib.insmos_debug_dw_kernel(2) == 0


def test_runner_scheduling_knobs_are_validated_on_the_host():
    """insmos_forward_regroup takes one decimal digit per 3D level (0 off, 1 / 2 / 3 = blocks of 256 / 1024 / 4096 rows, 4 = whole
    windows, 5 = parity class first; -1 = default) and insmos_forward_streams a 4-bit mask (-1 = default); anything else is
    refused.  Workspace sizes of the regrouping entry points are pure host arithmetic."""
    from insmos_amd import _lib
    lib = _lib.load()
    try:
        for ok in (0, 5, 3553, 4444, 1000, 5555, -1):
            assert lib.insmos_forward_regroup(ok) == 0, ok
        for bad in (6, 16, 3563, 9000, 55555, -2):
            assert lib.insmos_forward_regroup(bad) != 0, bad
        for ok in (0, 15, 2, -1):
            assert lib.insmos_forward_streams(ok) == 0, ok
        for bad in (16, -2, 99):
            assert lib.insmos_forward_streams(bad) != 0, bad
    finally:
        lib.insmos_forward_regroup(-1)
        lib.insmos_forward_streams(-1)
    assert lib.insmos_regroup_ws_bytes(0) == 0
    small, big = lib.insmos_regroup_ws_bytes(1000), lib.insmos_regroup_ws_bytes(300000)
    assert 16 * 1000 <= small < big and big >= 16 * 300000        # two key arrays + the sort's scratch
    # null / unsupported arguments never reach a launch
    assert lib.insmos_regroup_rows3d(None, 10, None, None, 4096, None, None, None, None, 0, None) != 0
    assert lib.insmos_regroup_rows3d(None, 0, None, None, 4096, None, None, None, None, 0, None) == 0      # nothing to do


def test_overlapped_reducer_contract_one_backward_per_reduce_and_close():
    """overlap=True: a second backward before reduce() would write into a bucket whose all-reduce is in flight -> it raises;
    close() removes the hooks (a second reducer over the same parameters then sees every gradient exactly once)."""
    import torch
    from insmos_amd.ddp import BucketedGradReducer
    params = {"a": torch.nn.Parameter(torch.ones(6)), "b": torch.nn.Parameter(torch.ones(3))}
    red = BucketedGradReducer(params, bucket_bytes=16, overlap=True)     # one bucket per parameter
    assert len(red.buckets) == 2
    loss = lambda: (params["a"] ** 2).sum() + (params["b"] ** 3).sum()   # noqa: E731
    loss().backward()
    assert red._launched == 2
    with pytest.raises(RuntimeError, match="arrived twice"):
        loss().backward()
    red.reset()                          # an aborted backward: without reset() every later backward would raise
    assert red._launched == 0 and not red._works
    for p in params.values():
        p.grad = None
    loss().backward()
    assert red._launched == 2
    red.reduce()
    g = params["a"].grad.clone()
    red.close()
    with pytest.raises(RuntimeError, match="after close"):
        red.reduce()
    red2 = BucketedGradReducer(params, bucket_bytes=16, overlap=True)
    for p in params.values():
        p.grad = None
    loss().backward()                   # only red2's hooks fire now
    assert red2._launched == 2 and red._launched == 0
    red2.reduce()
    assert torch.equal(params["a"].grad, g)
    red2.close()


def test_thread_local_precision_override_is_validated_and_per_trainer_context():
    from insmos_amd import _lib, autograd
    lib = _lib.load()
    for ok in (1, 3, 0, -1):
        assert lib.insmos_conv_precision_thread(ok) == 0
    for bad in (2, -2, 5):
        assert lib.insmos_conv_precision_thread(bad) != 0
    assert autograd.current_train_conv_precision() == 0
    with autograd.train_conv_precision(1):
        assert autograd.current_train_conv_precision() == 1
        with autograd.train_conv_precision(0):
            assert autograd.current_train_conv_precision() == 0
        assert autograd.current_train_conv_precision() == 1
    assert autograd.current_train_conv_precision() == 0
    with pytest.raises(ValueError):
        with autograd.train_conv_precision(3):
            pass


def test_forward_host_marks_entry_point_without_a_forward():
    """insmos_forward_host_marks (include/insmos_hip.h): the calling thread's last forward as "stage:us;..." text -- empty when no
    forward ran on this thread (or INSMOS_HOST_MARKS is off), INSMOS_EINVAL for a missing buffer; needs no GPU."""
    import ctypes
    from insmos_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(256)
    assert lib.insmos_forward_host_marks(buf, 256) == 0 and buf.value == b""
    assert lib.insmos_forward_host_marks(None, 256) == -1 and lib.insmos_forward_host_marks(buf, 0) == -1
    assert lib.insmos_bev_cosplit(-2) == -1 and lib.insmos_bev_cosplit(0) == 0 and lib.insmos_bev_cosplit(-1) == 0
    assert lib.insmos_bev_skip_ws_bytes(8, 125, 150) >= 8 * 150 * 10 * 4 and lib.insmos_bev_skip_ws_bytes(0, 125, 150) == 0


_FORCE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[2])
import bench
from insmos_amd.ddp import BucketedGradReducer
from insmos_amd.metrics import all_gather_confusion
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[1]
dist.init_process_group("gloo", rank=0, world_size=1)
cm = torch.arange(9, dtype=torch.int64).reshape(3, 3)
assert all_gather_confusion(cm) is cm                      # a world of one exchanges nothing ...
got = all_gather_confusion(cm, force=True)                 # ... unless asked to: the collective runs, the result is the input
assert got is not cm and torch.equal(got, cm)
params = {"a": torch.zeros(5, 3, requires_grad=True), "b": torch.zeros(7, requires_grad=True)}
for i, v in enumerate(params.values()):
    v.grad = torch.full_like(v, float(i + 1))
red = BucketedGradReducer(params, bucket_bytes=32)
assert red.reduce() == 2 and red.collectives_issued == 0   # default: no collective in a world of one
red = BucketedGradReducer(params, bucket_bytes=32, force_collective=True)
assert red.reduce() == 2 and red.collectives_issued == 2
assert torch.equal(params["a"].grad, torch.full((5, 3), 1.0)) and torch.equal(params["b"].grad, torch.full((7,), 2.0))

class Stub:                                                # bench.timed_steps is device-agnostic
    def __call__(self, batch, mode):
        return None, None, [torch.zeros(4, 3) for _ in batch]
class M:
    def compute_confusion_matrix(self, lg, gt, out=None):
        out[1, 1] += lg.shape[0]
dt, value, cm_all = bench.timed_steps(Stub(), [0, 1], [None, None], M(), 3, 1, 1, "cpu", lambda: None, force_collectives=True)
assert int(cm_all[1, 1]) == 3 * 2 * 4 and abs(value - 3 * 2 / dt) < 1e-9
dist.destroy_process_group()
print("OK")
"""


def test_forced_collectives_in_a_world_of_one_rank(tmp_path):
    """What tests/test_zz_gpu_rccl_world1.py runs over RCCL on the GPU box, over gloo here: `force` makes a one-rank group
    issue the N-rank code's collectives (metrics.all_gather_confusion, ddp.BucketedGradReducer, bench.timed_steps)."""
    script = tmp_path / "force_worker.py"
    script.write_text(_FORCE_WORKER)
    r = subprocess.run([sys.executable, str(script), str(29300 + os.getpid() % 150), ROOT], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=180)
    assert r.returncode == 0 and "OK" in r.stdout.decode(), r.stdout.decode()[-2000:]


_EMIT_WORKER = r"""
import ctypes, os, sys
sys.path.insert(0, sys.argv[1])
import bench
libc = ctypes.CDLL(None)
libc.printf(b"RCCL version : banner written through C stdio, still in libc's buffer on a pipe\n")
print("python-level chatter before the result")
bench.emit_result_line({"metric": "scans_per_sec", "value": 1.0})
libc.printf(b"a library flushing at teardown\n")      # must not reach the pipe
print("python-level chatter after the result")         # neither
"""


def test_bench_result_line_is_the_last_line_of_stdout(tmp_path):
    """What first contact with RCCL found on the GPU box: a library banner written through C stdio sat in libc's pipe buffer until
    exit and landed BEHIND the JSON line.  bench.emit_result_line flushes libc first, writes the line, and closes stdout behind it."""
    import json
    script = tmp_path / "emit_worker.py"
    script.write_text(_EMIT_WORKER)
    r = subprocess.run([sys.executable, str(script), ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().splitlines()
    assert json.loads(lines[-1]) == {"metric": "scans_per_sec", "value": 1.0}, lines
    assert any("banner" in l for l in lines[:-1]) and any("before the result" in l for l in lines[:-1])
    assert not any("teardown" in l or "after the result" in l for l in lines)


def test_bench_refuses_more_ranks_than_gpus_in_one_line():
    """First contact with a node that has fewer GPUs than --gpus: ONE clear line at once (no ranks spawned, no rendezvous timeout) --
    the function on counts, and the real command on this GPU-less host; a one-GPU dry run (--device-index) and gloo are let through."""
    import argparse
    import time
    import bench
    ns = argparse.Namespace(gpus=8, backend="nccl", device_index=None)
    with pytest.raises(SystemExit) as e:
        bench.check_enough_gpus(ns, device_count=1)
    assert "--gpus 8" in str(e.value) and "1 GPU(s)" in str(e.value) and "--device-index 0" in str(e.value)
    bench.check_enough_gpus(ns, device_count=8)
    bench.check_enough_gpus(argparse.Namespace(gpus=8, backend="gloo", device_index=None), device_count=0)
    bench.check_enough_gpus(argparse.Namespace(gpus=2, backend="nccl", device_index=0), device_count=1)
    bench.check_enough_gpus(argparse.Namespace(gpus=1, backend="nccl", device_index=None), device_count=0)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode()
    assert r.returncode != 0 and time.time() - t0 < 30.0, out[-1000:]
    assert "--gpus 8 but this node shows" in out and "torch.distributed.run" not in out, out[-1000:]


def test_bench_traffic_record_is_tied_to_the_binary(tmp_path):
    """roofline.traffic comes from a committed PMC pass: it is only printed when that pass was taken from the binary loaded now
    (lib_source_hash); another hash, or a record without one, reads as stale (traffic: null, traffic_stale: true)."""
    import argparse
    import json
    import bench
    args = argparse.Namespace(n_az=1886)
    rec = {"windows_per_launch": 8, "hbm_bytes_per_launch": 1.0, "hbm_bytes_per_window": 2.0, "lib_source_hash": "abcdef012345"}
    (tmp_path / "r06_pmc_traffic.json").write_text(json.dumps(rec))
    j, stale = bench.pmc_traffic(args, 8, "cfg2", profiles_dir=str(tmp_path), lib_hash="abcdef012345")
    assert j is not None and not stale and j["hbm_bytes_per_window"] == 2.0
    j, stale = bench.pmc_traffic(args, 8, "cfg2", profiles_dir=str(tmp_path), lib_hash="0123456789ab")
    assert j is not None and stale
    (tmp_path / "r07_pmc_traffic.json").write_text(json.dumps({k: v for k, v in rec.items() if k != "lib_source_hash"}))
    assert bench.pmc_traffic(args, 8, "cfg2", profiles_dir=str(tmp_path), lib_hash="abcdef012345")[1]      # latest record, no hash: stale
    assert bench.pmc_traffic(args, 4, "cfg2", profiles_dir=str(tmp_path), lib_hash="abcdef012345") == (None, False)   # other set size
    assert bench.pmc_traffic(args, 8, "cfg4", profiles_dir=str(tmp_path), lib_hash="abcdef012345") == (None, False)   # no cfg-4 pass
    (tmp_path / "r06_pmc_traffic_cfg4.json").write_text(json.dumps(dict(rec, windows_per_launch=2)))
    assert bench.pmc_traffic(args, 2, "cfg4", profiles_dir=str(tmp_path), lib_hash="abcdef012345")[1] is False
    h = bench.lib_source_hash()
    assert h is None or (len(h) == 12 and all(c in "0123456789abcdef" for c in h))


def test_gather_rows_backward_is_a_deterministic_scatter_add():
    """insmos_amd.autograd.gather_rows (the point -> voxel gathers of both heads, spconv_unet.py:408-410, motionnet.py:42-46): forward is
    plain row indexing; the backward -- a fixed-point int64 scatter-add, so that repeated rows add in any order to the same bits --
    equals torch's sort-based indexing backward to 1e-9 and gives identical bits for a permuted index stream (runs on the CPU: torch
    ops only)."""
    import torch
    from insmos_amd.autograd import gather_rows
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 3, generator=g, requires_grad=True)
    idx = torch.randint(0, 300, (20000,), generator=g)
    gy = torch.randn(20000, 3, generator=g) * 1e-3
    y = gather_rows(x, idx)
    assert torch.equal(y, x[idx])
    y.backward(gy)
    ref = torch.zeros(300, 3, dtype=torch.float64).index_add_(0, idx, gy.double())
    assert float((x.grad.double() - ref).abs().max()) < 1e-9
    perm = torch.randperm(20000, generator=g)
    x2 = x.detach().clone().requires_grad_(True)
    gather_rows(x2, idx[perm]).backward(gy[perm])
    assert torch.equal(x2.grad, x.grad)
