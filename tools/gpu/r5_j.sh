#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/dvq_logs
timeout 600 python train.py --base configs/stage1/dqvae-entropy-dual-r05_imagenet.yml --precision fp32x3 --max_steps 3 --steps_per_epoch 3 --val_batches 1 --logdir /tmp/dvq_logs data.params.batch_size=4 2>&1 | grep -v "Warn\|return get_obj\|amdgpu" | tail -4 | cut -c1-300
timeout 600 python train.py --base configs/stage2/uncond_imagenet_p6c18.yml --precision fp32x3 --max_steps 2 --steps_per_epoch 2 --val_batches 1 --logdir /tmp/dvq_logs data.params.batch_size=2 2>&1 | grep -v "Warn\|return get_obj\|amdgpu" | tail -4 | cut -c1-300
timeout 600 python train.py --base configs/stage1/dqvae-triple-r-03-03_imagenet.yml --max_steps 3 --steps_per_epoch 3 --val_batches 1 --logdir /tmp/dvq_logs data.params.batch_size=4 2>&1 | grep -v "Warn\|return get_obj\|amdgpu" | tail -3 | cut -c1-300
