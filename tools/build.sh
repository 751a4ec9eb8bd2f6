#!/bin/bash
# build libdvq_hip.so in-tree and load it; non-zero exit (and the compiler's message) when either fails -- use as `bash tools/build.sh && gpurun ...`
# Then the generated-code lint of the LDS-DMA kernels (tools/lint_dma_barriers.py; only re-assembles sources that changed; DVQ_BUILD_LINT=0 skips it).
cd "$(dirname "$0")/.."
if python -c "from dynamicvectorquantization_amd import build; build.build(); from dynamicvectorquantization_amd import _lib; print('libdvq_hip ABI', _lib.load().dvq_version())" > /tmp/dvq_build.log 2>&1; then
    tail -1 /tmp/dvq_build.log
else
    tail -25 /tmp/dvq_build.log
    exit 1
fi
if [ "${DVQ_BUILD_PROBES:-0}" = "1" ]; then
    # the -DDVQ_PROBES twin (timing experiments with wrong results, persistent conv_halo2): tools/debug/ only, DVQ_USE_PROBES_LIB=1
    python -m dynamicvectorquantization_amd.build --probes > /tmp/dvq_build_probes.log 2>&1 && echo "probe library built" || { tail -20 /tmp/dvq_build_probes.log; exit 1; }
fi
if [ "${DVQ_BUILD_LINT:-1}" != "0" ]; then
    if python tools/lint_dma_barriers.py > /tmp/dvq_lint.log 2>&1; then
        echo "dma-barrier lint ok ($(grep -c 'barriers,' /tmp/dvq_lint.log) kernels)"
    else
        grep -A3 "CHECK" /tmp/dvq_lint.log | head -30
        echo "dma-barrier lint FAILED (tools/lint_dma_barriers.py)"
        exit 1
    fi
fi
