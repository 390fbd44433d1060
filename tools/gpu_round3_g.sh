cd $GRAFT_REPO_ROOT
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for r in 1 2 3; do python tools/k1_alloc_probe.py | sed "s/.*K1 \([0-9.]*\) ms.*memset \([0-9.]*\).*/cold \1 \2/"; FIGH_CHAIN_HOTIN=1 python tools/k1_alloc_probe.py | sed "s/.*K1 \([0-9.]*\) ms.*memset \([0-9.]*\).*/hot \1 \2/"; done | paste - - - - - -
