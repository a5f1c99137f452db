"""Every kernel of the profiled steps from the PMC passes of tools/pmc_diag.sh (gpurun_out/pmcdiag_<tag>_<cfg>/):
   python tools/pmc_all_kernels.py C1 [tag] > profiles/rNN_all_kernels_pmc_C1.txt   (tag of the pmc_diag call, default r04_v2)"""
import re, ast, subprocess, sys
cfg=sys.argv[1]
tag=sys.argv[2] if len(sys.argv) > 2 else 'r04_v2'
out=subprocess.run(["python","tools/pmc_summary.py",f"gpurun_out/pmcdiag_{tag}_{cfg}","3"],capture_output=True,text=True).stdout
rows=[]
for l in out.splitlines():
    m=re.match(r'(.+?) (\{.*\})$', l.strip())
    if not m: continue
    d=ast.literal_eval(m.group(2)); n=m.group(1)
    if not d.get('SQ_WAVES') or not d.get('SQ_ACTIVE_INST_VALU'): continue
    rows.append((d['GRBM_GUI_ACTIVE'], n, d))
rows.sort(reverse=True)
print(f"# {cfg}: every kernel of the 3 profiled steps (2 timed + 1 warm-up) of bench.py, PMC passes of tools/pmc_diag.sh (build of the pmc_diag call),")
print("# sums over the 3 steps.  lanes = SQ_THREAD_CYCLES_VALU / (4 x SQ_ACTIVE_INST_VALU) (relative measure: k_match_pairs ~11-14 = its")
print("# ~45-55 of 64 lanes); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; valu/wave = SQ_INSTS_VALU / SQ_WAVES; sqc_miss = scalar data cache")
print(f"{'kernel':44s} {'waves':>9s} {'valu/wave':>9s} {'lanes':>6s} {'wait':>5s} {'sqc_miss':>8s} {'lds_conf':>8s}")
for _,n,d in rows:
    lanes=d['SQ_THREAD_CYCLES_VALU']/d['SQ_ACTIVE_INST_VALU']/4
    wait=d['SQ_WAIT_ANY']/max(d['SQ_WAVE_CYCLES'],1)
    sqc=d['SQC_DCACHE_MISSES']/max(d['SQC_DCACHE_REQ'],1)
    ldc=d['SQ_LDS_BANK_CONFLICT']/max(d['SQ_LDS_IDX_ACTIVE'],1) if d.get('SQ_LDS_IDX_ACTIVE') else 0
    print(f"{n[:44]:44s} {d['SQ_WAVES']:9d} {d['SQ_INSTS_VALU']/d['SQ_WAVES']:9.0f} {lanes:6.1f} {wait:5.2f} {sqc:8.3f} {ldc:8.3f}")
