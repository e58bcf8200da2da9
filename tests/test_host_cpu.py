"""CPU-only tests: host-side classes against the golden vectors captured from the reference, the C ABI
surface (library loads, exports every declared symbol; no compute without a GPU), loud failure of the
device path without a GPU, the layer-list / FLOP model, and the sharding + gather plumbing over gloo."""
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import nets_ref as N
from oracle import sampling_ref as S


def test_policy_bit_exact_vs_reference():
    from cgs_amd.sampling import PolicyAdaptive
    g = load_golden("g2_policy.npz")
    for m in ("sgd", "momentum", "ladam"):
        p = PolicyAdaptive(0.1, m)
        th = g["theta0"].copy()
        for i in range(3):
            r = p.apply_gradient(th, g["grads"][i], g["losses"][i])
            assert r is th                                        # in place on ndarrays (policy.py `theta -= ...`)
            np.testing.assert_array_equal(th, g[m][i])
        p.reset_moving_average()
        assert p.momentum is None and p.mean_square is None and p.loss is None
    p = PolicyAdaptive(0.5, "momentum")
    th = torch.from_numpy(g["theta0_map"])
    for i in range(3):
        th = p.apply_gradient(th, torch.from_numpy(g["grads_map"][i]))
        np.testing.assert_allclose(th.numpy(), g["momentum_map"][i], rtol=1e-6, atol=1e-7)
    with pytest.raises(NotImplementedError):
        PolicyAdaptive(0.1, "adamw").apply_gradient(np.zeros((2, 2)), np.zeros((2, 2)))
    with pytest.raises(ValueError):
        PolicyAdaptive(0.1, "ladam").apply_gradient(np.zeros((2, 2)), np.zeros((2, 2)))     # quirk Q7 made explicit


def test_refiner_cpu_config1_matches_reference():
    """BASELINE config 1 (Imbal-8Gaussians, B=512, K=10, ladam): the product's refiner_cpu.Refiner driven
    exactly like the reference's (duck-typed gan / sess / data), against the reference's own output."""
    from cgs_amd.sampling import refiner_cpu
    g = load_golden("g1_refiner_cpu.npz")
    Ws = [torch.from_numpy(w) for w in g["W"]]
    bs = [torch.from_numpy(b) for b in g["b"]]
    calls = []

    class Gan:
        fake_samples, fake_sigmoid, fake_saliency = "x", "sig", "sal"

    class Sess:
        def run(self, fetches, feed_dict):
            sig, sal = S.mlp_sigmoid_and_saliency(Ws, bs, feed_dict[Gan.fake_samples])
            calls.append(1)
            return [{"sig": sig, "sal": sal}[f] for f in fetches]

    class Data:
        def next_batch(self, n):
            return S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, n)

    args = types.SimpleNamespace(rollout_steps=10, rollout_rate=0.1, rollout_method="ladam")
    for mode in ("deterministic", "probabilistic"):
        ref = refiner_cpu.Refiner(args)
        ref.set_env(Gan, Sess(), Data())
        np.random.seed(2019)
        calls.clear()
        fake = g["fake"].copy()
        out = ref.manipulate_sample(fake, mode)
        np.testing.assert_array_equal(fake, g["fake"])            # input untouched
        assert len(calls) == int(g[mode + "_calls"][0]) == 12      # K+2 D evaluations
        assert str(out.dtype) == str(g[mode + "_dtype"][0])
        np.testing.assert_allclose(out, g[mode], rtol=0, atol=1e-6)
        assert ref.policy.momentum is None
    with pytest.raises(NotImplementedError):
        ref.manipulate_sample(g["fake"].copy(), "greedy")


def test_rejector_bit_exact_vs_reference():
    from cgs_amd.sampling import Rejector
    g = load_golden("g6_rejector.npz")
    for tag, pct in (("p60", 60.0), ("p100", 100.0), ("none", None)):
        r = Rejector()
        np.random.seed(2019)
        for c in range(3):
            samples = np.arange(257, dtype=np.float32).reshape(-1, 1)
            good = r.sampling(samples, g[f"{tag}_sig{c}"], shift_percent=pct)
            np.testing.assert_array_equal(r.last_accept, g[f"{tag}_mask{c}"])     # accept mask: bit-exact
            np.testing.assert_array_equal(good[:, 0], samples[g[f"{tag}_mask{c}"], 0])
            assert r.D_tilde_M == g[f"{tag}_M{c}"][0]
    r = Rejector()
    r.set_score_max(np.array(0.93, dtype=np.float32))
    assert r.D_tilde_M == g["set_score_max_M"][0]
    with pytest.raises(NotImplementedError):
        r.sampling(np.zeros((2, 1)), np.full((2, 1), 0.5), ranking=[0, 1])


def test_independence_sampler_exact_vs_reference():
    from cgs_amd.sampling import IndependenceSampler
    g = load_golden("g7_mh.npz")
    mh = IndependenceSampler(T=20)
    mh.set_score_curr(0.4)
    np.random.seed(2019)
    for c in range(2):
        good = mh.sampling(np.arange(400, dtype=np.float32).reshape(-1, 1), g[f"sig{c}"])
        assert good.dtype == np.float32
        np.testing.assert_array_equal(good[:, 0].astype(np.int64), g[f"accepted{c}"])
    with pytest.raises(AssertionError):
        mh.sampling(np.zeros((2, 1)), np.array([[0.5], [1.5]]))


def test_metrics_match_reference():
    from cgs_amd import metrics as Mx
    g = load_golden("g9_metrics.npz")
    thres = float(g["thres"][0])
    md, good = Mx.metrics_distance(g["model"], g["centeroids"], thres)
    assert abs(md - g["mean_dist"][0]) < 1e-12 and good == g["rate_good"][0]
    fv, fa = Mx.freq_category(g["model"], g["centeroids"], thres)
    np.testing.assert_allclose(fv, g["freqs_valid"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(fa, g["freqs_all"], rtol=0, atol=1e-15)
    assert abs(Mx.metrics_diversity(g["real"], g["model"], g["centeroids"], thres) - g["kl"][0]) < 1e-12
    assert abs(Mx.metrics_distribution(g["real"], g["model"], g["centeroids"], thres) - g["js"][0]) < 1e-12
    assert abs(Mx.metrics_diversity(g["real"], g["model"] + 100.0, g["centeroids"], thres) - g["far_kl"][0]) < 1e-12
    rs = np.random.RandomState(0)
    a = rs.randn(4000, 6)
    assert Mx.frechet_distance(a, a) < 1e-6
    assert abs(Mx.frechet_distance(a, a + 2.0) - 6 * 4.0) < 1e-6                 # pure mean shift: |d mu|^2


def test_toy_datasets_match_reference():
    from cgs_amd.datasets import ToyDataset, NoiseDataset
    g = load_golden("g8_toy.npz")
    for distr, ratio, B in (("Imbal-8Gaussians", 0.9, 512), ("8Gaussians", 0.5, 100), ("25Gaussians", 0.5, 60)):
        np.random.seed(2019)
        d = ToyDataset(distr=distr, scale=10.0, ratio=ratio)
        np.testing.assert_array_equal(d.next_batch(B), g[distr])
    np.random.seed(2019)
    np.testing.assert_array_equal(NoiseDataset().next_batch(33), g["noise"])
    assert abs(ToyDataset("Imbal-8Gaussians", 10.0, 0.9).std - 0.02 * 10.0 / 1.414) < 1e-15


def test_collaborate_fill_loop():
    """The accept/reject fill loop (nsgan/GAN.py:398-426) with a scripted proposer: exact bookkeeping."""
    from cgs_amd.evaluate import collaborate
    from cgs_amd.sampling import IndependenceSampler
    rs = np.random.RandomState(0)
    B = 50
    calls = []

    def propose():
        calls.append(1)
        return rs.randn(B, 3).astype(np.float32)

    score = lambda batch: rs.beta(2, 2, size=(len(batch), 1))
    np.random.seed(1)
    out, eff = collaborate(propose, score, IndependenceSampler(T=20), 30, 0.5)
    assert out.shape == (30, 3) and 0 < eff <= 1.0
    assert abs(eff - (lambda acc, prop: acc / prop)(round(eff * (30 + B * len(calls))), 30 + B * len(calls))) < 1e-12
    # efficiency cut-off: with min_efficiency = 0.9 the budget (eval_size / 0.9) is exhausted at once -> whole batches are taken
    calls.clear()
    out2, eff2 = collaborate(propose, score, IndependenceSampler(T=20), 120, 0.5, min_efficiency=0.9)
    assert len(calls) >= 2 and out2.shape == (120, 3)


class _ScriptedProposer:
    """A FusedProposer stand-in on the host (no GPU): images / sigmoids are fixed functions of z, so the one-batch-at-a-time loop and
    the fused rounds see the same proposals and the comparison isolates the loop's bookkeeping and its use of the global stream."""
    zdim = 5

    def __init__(self, b, G, depth=2):
        self.b, self.G, self.depth, self.slots, self.launched = b, G, depth, {}, 0

    @staticmethod
    def images(z):
        return np.tanh(z[:, :3]).astype(np.float32)

    @staticmethod
    def sigmoids(z):
        return (1.0 / (1.0 + np.exp(-z.sum(1, keepdims=True)))).astype(np.float32)

    def launch(self, z, refine=True):
        # like the device proposer, a slot's result buffers are REUSED in place by its next launch: a consumer that keeps a view of
        # them (the MH chain keeps the score it moved to) must have copied what it keeps
        k = self.launched % self.depth
        self.launched += 1
        if k not in self.slots:
            self.slots[k] = (np.empty((len(z), 3), np.float32), np.empty((len(z), 1), np.float32))
        self.slots[k][0][...] = self.images(z)
        self.slots[k][1][...] = self.sigmoids(z)
        return k

    def result(self, k):
        return self.slots[k]


@pytest.mark.parametrize("G,eval_size,min_eff", [(1, 40, None), (4, 100, None), (3, 64, 0.2), (8, 200, 0.5), (5, 37, 0.9)])
def test_collaborate_fused_equals_the_one_batch_loop(G, eval_size, min_eff):
    """collaborate_fused (G logical batches per device round, uniforms drawn ahead, chain run a round later) == collaborate fed one
    batch at a time (nsgan/GAN.py:398-426): the same accepted samples, the same efficiency, and the global numpy stream left in the
    same state -- including the efficiency cut-off branch, which stops consuming chain uniforms part-way (min_efficiency)."""
    from cgs_amd.evaluate import collaborate, collaborate_fused
    from cgs_amd.sampling import IndependenceSampler
    b = 16
    P = _ScriptedProposer(b, G)
    base_z = np.random.RandomState(3).uniform(-1, 1, (eval_size, P.zdim)).astype(np.float32)
    base = (P.images(base_z), P.sigmoids(base_z))
    last = {}

    def propose():
        last["z"] = np.random.uniform(-1, 1, [b, P.zdim]).astype(np.float32)
        return P.images(last["z"])

    def score(batch):
        return P.sigmoids(last["z"])
    np.random.seed(11)
    want, eff = collaborate(propose, score, IndependenceSampler(T=6), eval_size, 0.45, base=base, min_efficiency=min_eff)
    tail = np.random.uniform(size=3)
    np.random.seed(11)
    st = {}
    got, eff2 = collaborate_fused(P, IndependenceSampler(T=6), eval_size, 0.45, base=base, min_efficiency=min_eff, stats=st)
    np.testing.assert_array_equal(got, want)
    assert eff2 == eff
    np.testing.assert_array_equal(np.random.uniform(size=3), tail)          # the stream was rewound to where the reference's loop leaves it
    assert st["proposed"] % b == 0 and st["rounds"] >= 1 and st["discarded_batches"] >= 0


def test_independence_sampler_walk_with_predrawn_uniforms():
    """IndependenceSampler.walk(sigmoids, uniforms): uniforms drawn ahead from the global stream give the chain the per-proposal draws
    of idpsampler.py:50, bit for bit; float32 [B, 1] scores (what sess.run returns) and float64 ones; state type kept."""
    import copy
    from cgs_amd.sampling import IndependenceSampler
    rs = np.random.RandomState(2)
    for dt in (np.float32, np.float64):
        a = IndependenceSampler(T=5, B=1)
        a.set_score_curr(0.5)
        for it in range(3):
            x, sg = rs.randn(200, 2).astype(np.float32), rs.uniform(0.02, 0.98, (200, 1)).astype(dt)
            b = copy.deepcopy(a)
            np.random.seed(it)
            ya = a.sampling(x, sg)
            np.random.seed(it)
            yb = b.sampling(x, sg, uniforms=np.random.uniform(0, 1, size=200))
            np.testing.assert_array_equal(ya, yb)
            assert a.cnt_chain == b.cnt_chain and type(a.d_curr) is type(b.d_curr) and np.array_equal(a.d_curr, b.d_curr)
    fresh = IndependenceSampler()
    with pytest.raises(ValueError):
        fresh.walk(np.full((4, 1), 0.5), np.zeros(4))          # an unstarted chain takes its first proposal without a draw


def test_checkpoint_roundtrip_and_tf_name_cleaning(tmp_path):
    from cgs_amd import checkpoint as C
    P = {k: v.numpy() for k, v in N.init_params("mnist", 3, True).items()}
    for ext in ("safetensors", "npz"):
        path = str(tmp_path / f"m.{ext}")
        C.save(path, P)
        Q = C.load(path)
        assert set(Q) == set(P) and all(np.array_equal(P[k], Q[k]) for k in P)
        assert C.check_against_arch(Q, "mnist")
    tf_dump = {k + ":0": v for k, v in P.items()}
    tf_dump["discriminator/d_conv1/w/Adam:0"] = np.zeros(3); tf_dump["beta1_power:0"] = np.zeros(())
    assert set(C.clean_tf_names(tf_dump)) == set(P)
    bad = dict(P); bad.pop("generator/g_dc4/w")
    with pytest.raises(KeyError):
        C.check_against_arch(bad, "mnist")
    bad = dict(P); bad["generator/g_dc4/w"] = np.zeros((4, 4, 1, 32), np.float32)
    with pytest.raises(ValueError):
        C.check_against_arch(bad, "mnist")


def test_c_abi_library_exports_every_declared_symbol():
    from cgs_amd import lib
    header = open(os.path.join(ROOT, "include", "cgs_hip.h")).read()
    declared = set(re.findall(r"\b(cgs_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    l = lib.load()                                                  # dlopen + dlsym of each entry point
    for name in declared:
        assert getattr(l, name) is not None
    assert l.cgs_version() >= 100
    assert lib.conv_ws_bytes(lib.CONV_FWD, 5, 5, 2, 2, 64, 96) == 25 * 64 * 128 * 4     # (96 columns padded to 128)
    # a layer the split-bf16 form can serve (reduction over whole 32-channel chunks into a multiple of 128 channels): room for its
    # three bf16 planes, 6 bytes per weight, whatever the calling thread's contraction mode
    assert lib.conv_ws_bytes(lib.CONV_FWD, 5, 5, 2, 2, 64, 128) == 25 * 64 * 128 * 6
    assert lib.bn_ws_bytes(1000, 128) > 0


def test_the_library_is_tied_to_the_sources_it_was_built_from(tmp_path, monkeypatch):
    """the prebuilt .so is git-ignored and travels to the GPU box as a file, so nothing used to stop a stale binary from
    being measured under the hash of the sources lying next to it.  The Makefile embeds sha256(sources) into the library
    (``cgs_source_sha``, csrc/stamp.hip); ``lib.load`` compares it with the tree and refuses a mismatch.  Here: the in-tree library
    matches its tree; against a temp copy of the sources with one touched .hip file the check raises (and records ``stale`` instead
    under CGS_ALLOW_STALE / CGS_LIB); a binary-only install (no sources) skips the comparison."""
    import shutil
    from cgs_amd import lib
    l = lib.load()
    emb = l.cgs_source_sha().decode()
    assert re.fullmatch(r"[0-9a-f]{64}", emb) and emb == lib.source_hash() == lib.built_from() and lib.stale is False
    pkg = tmp_path / "pkg"
    shutil.copytree(os.path.join(ROOT, "collaborative-gan-sampling_amd", "csrc"), pkg / "csrc", ignore=shutil.ignore_patterns("*.o", ".source_sha"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    assert lib.source_hash(str(pkg)) == emb                            # the copy hashes like the tree ...
    lib._check_stamp(l, here=str(pkg))
    assert lib.stale is False
    with open(pkg / "csrc" / "elementwise.hip", "a") as f:             # ... until a kernel source moves on
        f.write("\n// touched\n")
    assert lib.source_hash(str(pkg)) != emb
    monkeypatch.delenv("CGS_LIB", raising=False)
    monkeypatch.delenv("CGS_ALLOW_STALE", raising=False)
    with pytest.raises(lib.StaleLibraryError, match="rebuild"):
        lib._check_stamp(l, here=str(pkg))
    assert lib.stale is True
    monkeypatch.setenv("CGS_ALLOW_STALE", "1")
    lib._check_stamp(l, here=str(pkg))                                  # tolerated on request, and recorded
    assert lib.stale is True
    # the Makefile of the touched copy would stamp the NEW hash: same file list, same byte order as lib.source_hash
    import subprocess
    sha = subprocess.run(["make", "-C", str(pkg / "csrc"), "-n", "stamp.o"], capture_output=True, text=True).stdout
    assert lib.source_hash(str(pkg)) in sha, sha[-400:]
    lib._check_stamp(l, here=str(tmp_path / "nothing_here"))            # binary-only install: nothing to compare with
    assert lib.stale is None
    monkeypatch.delenv("CGS_ALLOW_STALE")
    lib._check_stamp(l)
    assert lib.stale is False


def test_contraction_mode_is_per_thread_host_state():
    """cgs_set_contraction (include/cgs_hip.h): thread-local, validated, and the family query answers for the mode in force --
    host arithmetic only (no launch)."""
    import threading
    from cgs_amd import lib
    l = lib.load()
    assert lib.get_contraction() == "f32"
    big = (lib.CONV_FWD, 1024, 16, 16, 128, 0, 0, 256, 5, 5, 2, 2, lib.EPI_NONE, 1 << 26)        # dcgan64 d_h2 at the headline's batch
    small = (lib.CONV_FWD, 4, 16, 16, 128, 0, 0, 256, 5, 5, 2, 2, lib.EPI_NONE, 1 << 26)
    narrow = (lib.CONV_FWD, 1024, 32, 32, 64, 0, 0, 48, 5, 5, 2, 2, lib.EPI_NONE, 1 << 26)        # 48 output channels: no whole 64-column wave tile, never
    assert l.cgs_conv_family(*big) == lib.FAMILY_IGEMM
    try:
        assert lib.set_contraction("bx6") == "f32" and lib.get_contraction() == "bx6"
        assert l.cgs_conv_family(*big) == lib.FAMILY_IGEMM_BX6
        assert l.cgs_conv_family(*small) == lib.FAMILY_IGEMM                 # too small to gain: stays on the exact-fp32 kernel
        assert l.cgs_conv_family(*narrow) == lib.FAMILY_IGEMM
        seen = []
        t = threading.Thread(target=lambda: seen.append(int(l.cgs_get_contraction())))     # another host thread keeps its own default
        t.start(); t.join()
        assert seen == [lib.CONTRACTION_F32]
        lib.set_contraction("bx6_all")
        assert l.cgs_conv_family(*small) == lib.FAMILY_IGEMM_BX6 and l.cgs_conv_family(*narrow) == lib.FAMILY_IGEMM
        assert l.cgs_set_contraction(7) == lib.EINVAL and lib.get_contraction() == "bx6_all"
        # the fused-statistics queries answer for both implicit-GEMM families alike (same partial rows)
        assert l.cgs_conv_stat_partials(1024, 16, 16, 128, 256, 5, 5, 2, 2, 1 << 26) == 2 * (1024 * 64 // 128)
    finally:
        lib.set_contraction("f32")
    assert l.cgs_conv_stat_partials(1024, 16, 16, 128, 256, 5, 5, 2, 2, 1 << 26) == 2 * (1024 * 64 // 128)


def test_group_statistics_layout_is_host_arithmetic():
    """cgs_conv_stat_layout (include/cgs_hip.h): which partial rows of a *_fwd_stats call belong to a group of consecutive images follows
    from the launch's row order alone -- no GPU call -- for the image-major and the pixel-major (whole 128-image tiles) orders, the
    forward and the transposed direction; layouts a group does not own whole rows of are refused."""
    import ctypes as C
    from cgs_amd import lib
    l = lib.load()
    WS = 1 << 30

    def layout(op, B, H, Cin, Ho, Cout, k, s, grp):
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        rows = l.cgs_conv_stat_layout(op, B, H, H, Cin, Ho, Ho, Cout, k, k, s, s, grp, WS, C.addressof(a), C.addressof(b), C.addressof(c))
        return (rows, a.value, b.value, c.value)
    # PatchGAN conv at batch 8, one group per sample: rows (image, pixel): 256 pixels = 4 rows of 64 per sample, one segment
    assert layout(lib.CONV_FWD, 8, 32, 32, 0, 64, 4, 2, 1) == (32, 4, 1, 32)
    # transposed conv 16x16 -> 32x32: four parity classes of 256 pixels, a sample's 4 rows in each class, classes 32 rows apart
    assert layout(lib.DECONV_FWD, 8, 16, 64, 32, 32, 3, 2, 1) == (128, 4, 4, 32)
    # fused logical batches (4 x 64) of D's batch norm, pixel-major whole tiles: a row = 64 images of ONE of the 64 output pixels
    assert layout(lib.CONV_FWD, 256, 16, 32, 0, 64, 5, 2, 64) == (256, 1, 64, 4)
    # the whole batch as one group owns every partial row, whatever the row order (round 6: one segment of all rows -- also where a
    # proper grouping is refused, e.g. 9 pixels per sample)
    assert layout(lib.CONV_FWD, 256, 16, 32, 0, 64, 5, 2, 256) == (256, 256, 1, 0)
    assert layout(lib.CONV_FWD, 8, 6, 32, 0, 64, 3, 2, 8) == (2, 2, 1, 0)
    # round 6: the same question for the BACKWARD-DATA launches that leave a norm backward's column sums (cgs_*_bwd_data_nstats): the rows are
    # those of the gradient the launch writes -- conv 32x32x32 -> 16x16x64, stride 2: its backward-data runs four parity classes of 16x16 pixels
    # over the 32-channel input side, a sample's 4 rows in each class, classes 2 * ceil(8 * 256 / 128) = 32 rows apart
    assert layout(lib.CONV_BWD_DATA, 8, 32, 32, 0, 64, 4, 2, 1) == (128, 4, 4, 32)
    # ... a transposed conv's backward-data is a strided forward conv of the gradient: one class, 16 pixels of 256 images pixel-major
    assert layout(lib.DECONV_BWD_DATA, 256, 4, 64, 8, 32, 4, 2, 64) == (64, 1, 16, 4)
    assert layout(lib.CONV_BWD_DATA, 8, 64, 3, 0, 64, 5, 2, 8)[0] == 0           # a 3-channel gradient: another kernel family
    # refused: 9 pixels per sample; a logical batch of 32 under the pixel-major order; odd output (unequal parity classes); 3-channel family
    assert layout(lib.CONV_FWD, 8, 6, 32, 0, 64, 3, 2, 1)[0] == 0
    assert layout(lib.CONV_FWD, 256, 16, 32, 0, 64, 5, 2, 32)[0] == 0
    assert layout(lib.DECONV_FWD, 8, 5, 32, 9, 64, 3, 2, 8)[0] == 0
    assert layout(lib.CONV_FWD, 8, 32, 3, 0, 64, 5, 2, 1)[0] == 0


def test_device_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cgs_amd import kernels as K, lib
    from cgs_amd.engine import RefineEngine
    with pytest.raises(lib.CgsError):
        K.conv2d_fwd(torch.zeros(1, 4, 4, 8), torch.zeros(5, 5, 8, 8), None)      # CPU tensors are rejected, no fallback
    with pytest.raises(lib.CgsError):
        RefineEngine("mnist", {}, 4, device="cpu")


def test_layer_lists_agree_with_oracle_and_flop_model():
    from cgs_amd import nets
    for arch in ("mnist", "dcgan32", "dcgan64"):
        assert nets.param_shapes(arch) == {k: tuple(v) for k, v in N.param_shapes(arch).items()}
        assert nets.macs_per_sample(arch) == N.macs_per_sample(arch)
    # GFLOP per refined sample
    assert abs(nets.refine_flops_per_sample("mnist", 50) / 1e9 - 3.987) < 0.01
    assert abs(nets.refine_flops_per_sample("dcgan32", 20) / 1e9 - 5.631) < 0.01
    assert abs(nets.refine_flops_per_sample("dcgan64", 20) / 1e9 - 22.522) < 0.01


def _gloo_worker(rank, world, port, out):
    import torch.distributed as dist
    from cgs_amd import dist as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = torch.from_numpy(D.z_batches(rank, 3, 4, 5))

    def fake_refine(zb):                       # stand-in for the GPU engine: deterministic function of z
        return zb.view(4, 5, 1).repeat(1, 1, 2), zb.sum(1), torch.full((4,), float(rank))
    img, logit, step = D.refine_pool(fake_refine, z)
    rows = D.all_gather_floats([rank, 10.0 + rank])               # bench.py's bookkeeping gather (host tensors under gloo)
    mine = torch.zeros(world * 2, 3)
    got = D.gather_pool(torch.full((2, 3), float(rank)), out=mine)   # caller-owned pool buffer
    assert got is mine and all((mine[2 * r:2 * r + 2] == r).all() for r in range(world))
    if rank == 0:
        torch.save((img, logit, step, torch.from_numpy(rows)), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharding_and_gather_over_gloo(tmp_path, world):
    """SURVEY 8e on CPU: one process per rank, rank-offset z shards, one all-gather of the pool, rank-major rows -- at world 2 and at the
    world size of BASELINE configs 4 / 5 (8)."""
    import torch.multiprocessing as mp
    from cgs_amd import dist as D
    out = str(tmp_path / "pool.pt")
    import socket
    with socket.socket() as so:                                   # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_gloo_worker, args=(world, port, out), nprocs=world, join=True)
    img, logit, step, rows = torch.load(out)
    assert rows.shape == (world, 2) and rows[:, 0].tolist() == [float(r) for r in range(world)] and rows[:, 1].tolist() == [10.0 + r for r in range(world)]
    assert img.shape == (world * 3 * 4, 5, 2) and logit.shape == (world * 12,) and step.shape == (world * 12,)
    for r in range(world):
        z = torch.from_numpy(D.z_batches(r, 3, 4, 5)).reshape(12, 5)
        np.testing.assert_array_equal(img[r * 12:(r + 1) * 12, :, 0].numpy(), z.numpy())       # rank-major pool
        np.testing.assert_allclose(logit[r * 12:(r + 1) * 12].numpy(), z.sum(1).numpy(), rtol=1e-6)
        assert (step[r * 12:(r + 1) * 12] == r).all()
    assert not np.array_equal(D.z_batches(0, 1, 4, 5), D.z_batches(1, 1, 4, 5))                  # disjoint shards
    assert D.batch_owner(5, 4) == (1, 1) and D.rank_seed(3) == 2022
    assert D.gather_pool(torch.ones(2, 3)).shape == (2, 3)                                        # no group: identity


def test_bench_self_launches_its_ranks_as_child_processes(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): the parent must start torch.distributed.run as a CHILD process
    with the same arguments, on the loopback rendezvous, before touching the GPU, and exit with the child's code."""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"])
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before the launch")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_final_line_stays_small_and_carries_the_contract(tmp_path, capsys):
    """BENCH_r05.json parsed as null because the one stdout line had grown to 22 kB (the driver keeps 8 kB of tail).
    ``bench.compact_line`` builds the line from the full record; here from a canned one -- round 5's 22 kB line, the largest record a
    default run has produced, plus a `dist` object as an N > 1 run adds -- it must stay under ``LINE_BUDGET`` (6 kB), keep every
    contract key, the dominant kernel's roofline and the cpu_baseline, summarise every extra in one number, and ``emit`` must write
    the full record to the sidecar the line names."""
    import importlib
    import json
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    canned = os.path.join(ROOT, "profiles", "r05_ag_final_default_hipgraph_bench.log")
    full = json.loads([l for l in open(canned) if l.startswith("{")][-1])
    assert len(json.dumps(full)) > 20000                                   # (the record that did not parse)
    full["dist"] = {"backend": "nccl", "world_size": 8, "ranks_seen": 8, "devices": "one per rank", "pool_bytes": 402653184, "pool_buffers_per_rank": 2,
                    "pool_bytes_per_rank_total": 805306368, "device_mem_free_before_pools": 300000000000, "device_mem_total": 309220868096,
                    "gathers_per_step": 1, "gather_ms_per_step": 2.3012, "gather": "all_gather_into_tensor on the step's stream, HIP events around it",
                    "per_rank_samples_per_s": [6501.2, 6733.9], "hipgraph_ranks": 8, "pool_rows_match_ranks": True, "pool_rank_sums_distinct": True,
                    "rank0_profile_step_wall_s": 1.234}
    full["config"]["hipgraph_fallback"] = "RuntimeError: " + "x" * 400
    line = bench.compact_line(full, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_BUDGET == 6000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "lib", "dist", "summary", "detail"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and "workload" in line["config"]
    r = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "nominal_frac", "traffic", "algorithmic_bytes", "traffic_over_algorithmic",
              "launches", "avg_launch_us", "share_of_step", "step_executed_frac"):
        assert r[k] == full["roofline"][k], k
    assert "note" not in r and "timing" not in r
    c = line["cpu_baseline"]
    assert set(c) == {"value", "unit", "cores", "kind", "sample"} and len(c["sample"]) <= 120 and c["value"] == full["cpu_baseline"]["value"]
    s = line["summary"]
    assert s["bx6"]["samples_per_s"] == full["bx6"]["value"]
    assert set(s["other_configs"]) == {"mnist", "dcgan32", "cyclegan256", "synthetic2d"}
    for a in ("mnist", "dcgan32", "cyclegan256"):
        assert s["other_configs"][a]["samples_per_s"] == full["other_configs"][a]["samples_per_s"]
        assert s["other_configs"][a]["step_executed_frac"] == full["other_configs"][a]["roofline"]["step_executed_frac"]
    assert s["class_surface"]["mnist"] == {k: full["class_surface"]["mnist"][k]["samples_per_s"] for k in ("engine", "generic", "fused")}
    assert s["f1"]["accepted_samples_per_s"] == full["f1"]["accepted_samples_per_s"]
    assert s["shaping_iteration_ms"] == {"mnist": full["shaping"]["mnist"]["iteration_ms"], "dcgan64": full["shaping"]["dcgan64"]["iteration_ms"]}
    # emit: ONE stdout line (the compact one), the full record in the sidecar
    side = tmp_path / "detail.json"
    bench.emit(full, str(side))
    out = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(out) == 1 and len(out[0]) < bench.LINE_BUDGET and json.loads(out[0])["value"] == full["value"]
    assert json.loads(side.read_text())["other_configs"]["mnist"]["roofline"] == full["other_configs"]["mnist"]["roofline"]
    # a tree that cannot be written costs the sidecar, never the line
    bench.emit(full, str(tmp_path / "no_such_dir" / "detail.json"))
    cap = capsys.readouterr()
    assert "detail" not in json.loads(cap.out.strip()) and "could not write" in cap.err


def test_tf_checkpoint_converter_runs_against_the_checkpoint_reader_interface(tmp_path, monkeypatch):
    """tools/tf_ckpt_to_safetensors.py (reference saver: nsgan/GAN.py:465-491) has no TensorFlow to run against in
    this image.  Its own logic -- argument handling, reading every variable through the ``tf.train.load_checkpoint`` /
    CheckpointReader interface (``get_variable_to_shape_map``, ``get_tensor``), TF-name cleaning (':0' suffixes, optimizer
    slots, ``beta1_power``), validation against the arch, the .safetensors it writes -- executes here against a reader with
    that interface over an in-memory dump laid out like a TF1 Saver checkpoint of the mnist net."""
    import importlib.util
    import sys
    import types
    from cgs_amd import checkpoint
    from oracle import nets_ref as N
    P = N.init_params("mnist", 2019, True)
    dump = {k: v.numpy() for k, v in P.items()}
    dump["discriminator/d_conv1/w/Adam"] = np.zeros_like(dump["discriminator/d_conv1/w"])       # optimizer slots the Saver also writes
    dump["discriminator/d_conv1/w/Adam_1"] = np.zeros_like(dump["discriminator/d_conv1/w"])
    dump["beta1_power"] = np.float32(0.5)
    dump["beta2_power"] = np.float32(0.999)
    seen = {}

    class Reader:
        def get_variable_to_shape_map(self):
            return {k: list(np.shape(v)) for k, v in dump.items()}

        def get_tensor(self, name):
            return dump[name]

    def load_checkpoint(path):
        seen["path"] = path
        return Reader()
    tf = types.ModuleType("tensorflow")
    tf.train = types.SimpleNamespace(load_checkpoint=load_checkpoint)
    monkeypatch.setitem(sys.modules, "tensorflow", tf)
    spec = importlib.util.spec_from_file_location("tf_ckpt_to_safetensors", os.path.join(ROOT, "tools", "tf_ckpt_to_safetensors.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    out = str(tmp_path / "mnist_5000.safetensors")
    monkeypatch.setattr(sys, "argv", ["tf_ckpt_to_safetensors.py", "checkpoint/GAN_mnist_64_62/GAN/model-5000", out, "--arch", "mnist"])
    tool.main()
    assert seen["path"].endswith("model-5000")
    loaded = checkpoint.load(out)
    assert set(loaded) == set(P)                                         # slots and power accumulators dropped, every variable kept
    for k, v in P.items():
        assert torch.equal(torch.as_tensor(loaded[k]), v), k
    monkeypatch.setattr(sys, "argv", ["tf_ckpt_to_safetensors.py", "x/model-1", out, "--arch", "dcgan32"])
    with pytest.raises(KeyError):                                        # a checkpoint of another net is refused, not written
        tool.main()


def test_refiner_unwraps_the_reference_wiring_without_a_gpu():
    """collaborator.Refiner._unwrap (host logic of the engine detection): functools.partial objects -- also nested -- of
    BOUND methods with keyword arguments only resolve to (owner, function, keywords); positional arguments, plain functions,
    lambdas and unexpected keywords do not.  (The GPU suite checks the whole detection against a real model.)"""
    from functools import partial
    from cgs_amd.sampling.collaborator import Refiner

    class M:
        def discriminator(self, x, is_training=True, reuse=False):
            return x

        def feature_to_data(self, f, is_training=False):
            return f
    m = M()
    allowed = ("is_training", "reuse")
    assert Refiner._unwrap(m.discriminator, allowed) == (m, M.discriminator, {})
    assert Refiner._unwrap(partial(m.discriminator, is_training=True, reuse=True), allowed) == (m, M.discriminator, {"is_training": True, "reuse": True})
    # nested partials: the OUTER keyword wins, as functools applies them
    assert Refiner._unwrap(partial(partial(m.discriminator, is_training=False), is_training=True), allowed)[2] == {"is_training": True}
    assert Refiner._unwrap(partial(m.discriminator, 1.0), allowed) is None                    # positional argument bound
    assert Refiner._unwrap(partial(m.discriminator, name="d"), allowed) is None               # a keyword the engine does not know
    assert Refiner._unwrap(lambda x: m.discriminator(x), allowed) is None                     # opaque callable
    assert Refiner._unwrap(M.discriminator, allowed) is None                                  # unbound function
    assert Refiner._unwrap(partial(m.feature_to_data, is_training=False), ("is_training",)) == (m, M.feature_to_data, {"is_training": False})
    r = Refiner(3, 0.1)
    assert r.use_graph is True and r.path is None and r.graph_fallback is None               # the class-surface defaults (hipGraph on)


def test_contraction_kernels_keep_their_accumulators_in_registers():
    """A correct-but-4x-slower build is invisible to the parity tests: when an accumulator array is indexed at run time (an epilogue
    loop the compiler did not unroll) it moves to scratch memory.  Compile the two implicit-GEMM sources for gfx950 with the resource
    remarks on and require: no scratch, no spilled vector registers in any igemm kernel (hipcc cross-compiles without a GPU)."""
    import subprocess
    csrc = os.path.join(ROOT, "collaborative-gan-sampling_amd", "csrc")
    for src in ("igemm_bx6.hip", "igemm.hip"):
        out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                              "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(csrc, src), "-o", os.devnull],
                             capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0, out.stderr[-2000:]
        name, seen = None, 0
        for line in out.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line)
            if m and name and "igemm" in name and "pack" not in name and "reduce" not in name:
                seen += 1
                assert int(m.group(2)) == 0, f"{src}: {name}: {m.group(1)} = {m.group(2)}"
        assert seen >= 8, (src, seen)
