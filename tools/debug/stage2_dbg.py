import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from dynamicvectorquantization_amd import synth, runtime as rt
from dynamicvectorquantization_amd.config import instantiate_from_config
from test_gpu_stage2 import dualformer_config
dev = torch.device("cuda:0")
with rt.compute_dtype_ctx(torch.bfloat16):
    torch.manual_seed(0)
    model = instantiate_from_config(dualformer_config()).to(dev).eval()
    fs = model.first_stage_model
    x = torch.from_numpy(synth.half_flat_images(4, 64, seed=11)).to(dev)
    with torch.no_grad():
        rec1 = fs(x)[0]
        codes1 = fs._last["codes"].clone(); q1 = fs._last["quant"].float().clone()
        enc = fs.encode(x)
        quant, info, grain = enc[0], enc[2], enc[3]
        codes2 = info[2]
        print("codes equal:", bool((codes1.reshape(-1) == codes2.reshape(-1)).all()), int((codes1.reshape(-1) != codes2.reshape(-1)).sum()))
        print("quant diff:", float((quant.permute(0,2,3,1) - q1).abs().max()))
        z = model.permuter(indices=codes2, grain_indices=grain)
        back = model.permuter.forward_back(z["coarse_content"], z["fine_content"], z["coarse_position"], z["fine_position"])
        print("perm round trip:", bool((back == codes2).all()), int((back != codes2).sum()))
        rec_a = fs.decode(quant)
        print("decode(quant) vs fwd:", float((rec_a - rec1).abs().max()))
        q3 = fs.get_code_emb_with_depth(codes2)
        print("emb vs quant:", float((q3.permute(0,3,1,2) - quant).abs().max()))
        rec_b = fs.decode(q3.permute(0,3,1,2))
        print("decode(emb) vs fwd:", float((rec_b - rec1).abs().max()))
