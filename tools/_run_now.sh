export K="row_blocks"
export VARIANTS="tree tree@SVX_EDIT_BLOCKED=1 tree@SVX_EDIT_BLOCKED=1,SVX_EDIT_BLOCKED_K=0 tree@SVX_EDIT_BLOCKED=1,SVX_EDIT_BLOCKED_WALK=1500 tree@SVX_EDIT_BLOCKED=1,SVX_EDIT_BLOCKED_K=0,SVX_EDIT_BLOCKED_WALK=4000" WL="c1"
bash tools/gpu_r06.sh quick r06L
tail -5 gpurun_out/r06L_pytest_quick.txt
