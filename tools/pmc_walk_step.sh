export MFA_WALK=table MFA_REGIONS=0 STRINGS=65536 KERNEL=walk_kernel PMC_SETS="SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
bash tools/pmc_one.sh 1 4096 4096 > gpurun_out/pmc_walk.txt 2>&1
python3 - <<'PY'
d={}
for l in open('gpurun_out/pmc_walk.txt'):
    f=l.split()
    if len(f)==2 and f[0].startswith('SQ_'): d[f[0]]=float(f[1])
it=65536*4097/64
print("per wave-iteration: VALU %.0f SALU %.0f branch %.0f LDS %.0f SMEM %.0f | cycles %.0f wait_any %.0f%% active %.0f%%" % (d['SQ_INSTS_VALU']/it, d['SQ_INSTS_SALU']/it, d['SQ_INSTS_BRANCH']/it, d['SQ_INSTS_LDS']/it, d['SQ_INSTS_SMEM']/it, 4*d['SQ_WAVE_CYCLES']/it, 100*d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES'], 100*d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVE_CYCLES']))
PY
