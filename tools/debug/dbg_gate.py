import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_kernels as tk
dev = torch.device("cuda:0")
for case in tk.PIPE_CASES:
    try:
        tk.test_conv_pipelined_igemm_kernel(dev, case, "lrelu+gate")
        print(case, "ok")
    except AssertionError as e:
        print(case, "FAIL", str(e)[:200].replace("\n", " "))
