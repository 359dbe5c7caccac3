export K="routes_do_not_change"
export VARIANTS="tree tree@SVX_EDIT_SPLIT_CLS=7 tree@SVX_EDIT_SPLIT_CLS=5" WL="c1 c2 c4"
bash tools/gpu_r06.sh quick r06W
