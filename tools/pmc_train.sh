export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$n -- python tools/bench_train.py 32 1 > /tmp/pmc.log 2>&1
done
python - <<'PY'
import glob, pandas as pd
fr=[]
for f in glob.glob('/tmp/pmc_*/*/*counter_collection.csv')+glob.glob('/tmp/pmc_*/*counter_collection.csv'):
    t=pd.read_csv(f); fr.append(t)
t=pd.concat(fr)
t=t[t.Kernel_Name.str.contains('gemm_f32')]
t['k']=t.Kernel_Name.str.extract(r'gemm_f32_kernel<([^>]*)>')[0]
g=t.groupby(['k','Counter_Name'])['Counter_Value'].sum().unstack(1)
pd.set_option('display.width',250)
print(g.to_string())
PY
