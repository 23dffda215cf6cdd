# usage: bash tools/r5_run.sh TAG -- the GPU-side command file of one gpurun call (edited per call; see git history)
