# start-up delay of part of the workgroups (ablation build, FIGH_WY_DELAY=mode,cycles): are the two workgroups of a CU
# better off out of phase?  chained human pass (keeps the blocks: no natural de-phasing by the zero fill) and dense matrices
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
bash tools/chain_ab.sh >/dev/null 2>&1   # (writes /tmp/chain_ab.py)
for d in "0,0" "1,10000" "2,10000" "3,10000" "4,20000" "4,40000" "1,5000"; do
  echo "human chained delay=$d: $(FIGH_WY_DELAY=$d timeout 300 python /tmp/chain_ab.py 2>&1 | tail -1 | cut -c1-120)"
done
for n in 331 191; do
  for d in "0,0" "1,10000" "2,10000" "4,20000" "4,40000"; do
    echo "dense n=$n delay=$d $(FIGH_WY_DELAY=$d timeout 200 python tools/wide_tsqr_bench.py 4e6 $n 2>&1 | grep level0 | sed 's/.*| level0/level0/; s/| merges.*//')"
  done
done
