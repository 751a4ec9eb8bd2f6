"""Two data-parallel ranks with the real kernels on ONE GPU (gloo moves the device tensors; RCCL needs one device per rank):
the two-rank run on half batches must reproduce the single-process run on the whole batch.  `pytest -m gpu`."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_two_rank_step_equals_single_process_on_the_whole_batch(dev, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("DVQ_FORCE_DP", None)
    port = 29700 + (os.getpid() % 200)
    outs = {}
    for mode in ("dp", "single"):
        out = str(tmp_path / f"{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dp2_gloo_check.py"), mode, str(port), out], env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "DP2_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs[mode] = np.load(out)
    dp, one = outs["dp"], outs["single"]
    # rank 0 logs the loss of ITS half batch, the single process the loss of the whole batch: compare what is rank-independent --
    # the state the step leaves behind.  fp32 kernels, same math in another summation order; Adam turns the sign noise of
    # (near) zero gradients into +-lr steps (same bound as test_step_graph_matches_eager[ae]).
    steps, lr_max, tp = dp["losses"].shape[0], 2e-4, 2e-3
    assert np.isfinite(dp["losses"]).all() and np.isfinite(one["losses"]).all()
    n_checked = 0
    for k in one.files:
        if not k.startswith("p:"):
            continue
        a, b = dp[k].astype(np.float64), one[k].astype(np.float64)
        assert np.linalg.norm(a - b) <= tp * np.linalg.norm(b) + 0.05 * lr_max * steps * a.size ** 0.5, k
        n_checked += 1
    assert n_checked > 100
    for k in ("b:quantize.codebook.cluster_size_ema", "b:quantize.codebook.embed_ema"):
        a, b = dp[k].astype(np.float64), one[k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 5e-2 * np.linalg.norm(b) + 1e-6, k
    a, b = dp["adam_m"].astype(np.float64), one["adam_m"].astype(np.float64)
    assert np.linalg.norm(a - b) <= 10 * tp * np.linalg.norm(b) + 1e-9
    # and the halves really differ from the whole: the two runs saw different batches per process
    assert abs(dp["losses"][0, 0] - one["losses"][0, 0]) > 0


def test_two_rank_complete_objective_keeps_replicas_identical(dev, tmp_path):
    """both optimizers (autoencoder with LPIPS + adaptive GAN weight, then the PatchGAN) under two ranks: every bucket of both
    gradient buffers exchanged, recorded segments replayed as launch lists between the collectives, and after 7 steps the two
    replicas' parameters and EMA buffers have equal checksums (asserted inside the ranks)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("DVQ_FORCE_DP", None)
    port = 29900 + (os.getpid() % 90)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dp2_gloo_check.py"), "dp_full", str(port), str(tmp_path / "full.npz")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DP2_OK" in r.stdout and "DP2_RANK_OK 1" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_rank_stage2_step_equals_single_process(dev, tmp_path):
    """StackGPT training under two ranks: the per-block gradient exchange launched from inside the backward, with that block's Linear
    weight gradients still on the side stream (joined before the range is pre-divided): same parameters as one process on the whole
    batch, same losses (both ranks see the same images, so every rank's loss is the whole-batch loss)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("DVQ_FORCE_DP", None)
    port = 29800 + (os.getpid() % 90)
    outs = {}
    for mode in ("dp_s2", "single_s2"):
        out = str(tmp_path / f"{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dp2_gloo_check.py"), mode, str(port), out], env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "DP2_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs[mode] = np.load(out)
    dp, one = outs["dp_s2"], outs["single_s2"]
    # same arithmetic up to the summation order of the atomically accumulated weight gradients: measured agreement 5e-7 over all five
    # steps (a 2.5e-5 difference at step 0 was how the LDS-DMA barrier race of the D = 64 code search showed itself)
    np.testing.assert_allclose(dp["losses"], one["losses"], rtol=2e-5)
    assert dp["losses"][-1] < dp["losses"][0]
    steps, lr, n_checked = dp["losses"].shape[0], 1e-3, 0
    for k in one.files:
        if k.startswith("p:"):
            a, b = dp[k].astype(np.float64), one[k].astype(np.float64)
            assert np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(b) + 0.05 * lr * steps * a.size ** 0.5, k
            n_checked += 1
    assert n_checked > 40


@pytest.mark.gpu
def test_vq_argmin_exact_while_gpu_is_shared(dev, tmp_path):
    """two processes and a bandwidth-hungry copy stream share the GPU: every code search (fp32 and bf16 rows, D = 64 / 128 / 256,
    a ragged row count, an exactly tied pair of codes) still returns the oracle's exact argmin.  Regression test of the LDS-DMA
    barrier race (csrc/dvq_common.h dvq_dma_barrier, tools/lint_dma_barriers.py)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dp2_gloo_check.py"), "vq_share", "0", str(tmp_path / "x")],
                       env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP2_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("0 wrong indices") == 2, r.stdout[-1000:]


def test_two_rank_rccl_on_two_devices(dev, tmp_path):
    """the SAME two-rank checks over the real `nccl` (= RCCL) backend, one device per rank -- runs whenever the box shows >= 2 GPUs
    (the builder's boxes have one: skipped there; the first multi-GPU box exercises RCCL without anyone remembering to).
    Reference: /root/reference/train.py:227-230 (DDPPlugin), quantize2_mask.py:86-88,99-100 (all-reduce + broadcast)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DVQ_DP2_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("DVQ_FORCE_DP", None)
    port = 29600 + (os.getpid() % 90)
    outs = {}
    for mode in ("dp", "single", "dp_full"):
        out = str(tmp_path / f"{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dp2_gloo_check.py"), mode, str(port), out], env=env,
                           capture_output=True, text=True, timeout=900)
        if mode != "single" and "DP2_PG_OK" not in r.stdout:
            # the process group itself did not come up (driver / IPC / topology of this box): not a statement about this repository
            pytest.skip("RCCL two-rank group could not be created here: " + (r.stderr.strip().splitlines() or ["?"])[-1][:300])
        assert r.returncode == 0 and "DP2_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        if mode != "dp_full":
            outs[mode] = np.load(out)
    dp, one = outs["dp"], outs["single"]
    n_checked = 0
    for k in one.files:
        if k.startswith("p:"):
            a, b = dp[k].astype(np.float64), one[k].astype(np.float64)
            assert np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(b) + 0.05 * 2e-4 * dp["losses"].shape[0] * a.size ** 0.5, k
            n_checked += 1
    assert n_checked > 100
