#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; TAG=${TAG:-r05_h}
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider --tb=short -x -k "validation or checkpoint" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -n 5 gpurun_out/${TAG}_pytest.log | cut -c1-300
rm -rf /tmp/dvq_logs
timeout 600 python train.py --base configs/stage1/dqvae-entropy-dual-r05_imagenet.yml --max_steps 4 --steps_per_epoch 2 --val_batches 2 --logdir /tmp/dvq_logs data.params.batch_size=4 > gpurun_out/${TAG}_train.log 2>&1; echo "train.py exit $?"; grep -v "Warn\|return get_obj" gpurun_out/${TAG}_train.log | tail -8 | cut -c1-400; find /tmp/dvq_logs -name "*.ckpt" | sed 's|.*/||'
