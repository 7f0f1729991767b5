cd $GRAFT_REPO_ROOT
for lib in "" build/variants/libkpl_old.so; do
  tag=new; [ -n "$lib" ] && tag=old && export KPL_LIB_PATH=$PWD/$lib
  for r in 6 10; do
    echo "== $tag rmul=$r"
    bash tools/pmc_sorted.sh ${tag}$r "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" rmul=$r
    bash tools/pmc_sorted.sh ${tag}${r}b "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" rmul=$r
  done
done
