export K="c1_full or resident or radix or sampling"
export VARIANTS="tree tree@SVX_PREPACK_PRIO=normal" WL="c1 c2"
bash tools/gpu_r06.sh quick r06F
