"""Generated-code check (CPU, needs hipcc only): no kernel that stages operands by LDS-DMA reaches an s_barrier with DMA pieces of its
own wave unwaited, except the barriers listed -- with the reason -- in tools/lint_dma_barriers.py.  See csrc/dvq_common.h
(dvq_dma_barrier) for why __syncthreads() alone does not give that on gfx950."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_barrier_with_unwaited_lds_dma():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "lint_dma_barriers.py")], capture_output=True, text=True, timeout=1500)
    flagged = [l for l in r.stdout.splitlines() if "[CHECK]" in l]
    assert r.returncode == 0 and not flagged, "\n".join(flagged) + r.stderr[-2000:]
    checked = [l for l in r.stdout.splitlines() if "barriers," in l]
    assert len(checked) >= 20, len(checked)          # every DMA kernel of vq / igemm / conv_halo / conv_halo2 was walked
