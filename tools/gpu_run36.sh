cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2ab
EVO_AMD_LIBNAME=libevo_grprof.so EVO_AMD_NO_REBUILD=1 python tools/gemm_stage_profile.py 2>&1 | grep "^M=" | tee gpurun_out/r2ab/stages.log
