import json, re, sys, os
TAG=sys.argv[1]   # r06_mid / r06_final
SRC=sys.argv[2]   # directory holding the evidence files
ROOT='/root/repo'
d=json.load(open(f'{SRC}/{TAG}_bench_line.json')); r=d['roofline']
def line(n): return json.load(open(f'{SRC}/{TAG}_{n}.json'))
b=[line(f'b{i}_vae_line')['ms_per_step'] for i in (1,2,3,4)]
b1=line('b1_line')['ms_per_step']; c0=line('cfg0_line')['ms_per_step']
mf=json.load(open(f'{SRC}/{TAG}_pmc_MFMA.json'))['gemm']
busy=mf['SQ_VALU_MFMA_BUSY_CYCLES']['per_launch']/(mf['GRBM_GUI_ACTIVE']['per_launch']/8*1024)
log=open(f'{SRC}/{TAG}_gpu_pytest_measured.log').read()
npass=re.search(r'(\d+) passed', log).group(1)
smoke=re.search(r'smoke ok: (.*)', log).group(1)
cb=d['cpu_baseline']
traffic=r.get('traffic')
rep={
 '@@HEADLINE@@': f"**{d['value']:.2f} songs/s, RTF {d['rtf']:.0f}, {d['ms_per_step']:.1f} ms per 8-song pass** (box probe {d['box_probe']['mfma_random_bf16_tflops']:.0f} TFLOP/s pure MFMA; {d['gpu_state_under_load']['sclk_mhz']} MHz / {d['gpu_state_under_load']['package_power_w']:.0f} W under load: power-capped; BENCH_r05 on the driver's box: 470.9 ms)",
 '@@GEMM@@': f"{r['achieved']:.0f} TFLOP/s in-app = **{r['frac']:.3f} of the 2.5 PFLOP/s dense bf16 peak** ({d['box_probe']['gemm_achieved_over_probe']:.2f} of the box's pure-MFMA rate), {r['gemm_ms_per_pass']:.1f} ms of the pass, {r['avg_launch_us']:.1f} us per launch; MFMA busy {100*busy:.0f} %; fabric-side traffic {traffic/1e6 if traffic else float('nan'):.0f} MB per launch (2.3-2.5 x algorithmic)",
 '@@ATTNVAE@@': f"attention {r['attn_tflops']:.0f} TFLOP/s, {r['attn_ms_per_pass']:.1f} ms per pass (0.{int(r['attn_tflops']/2500*100):02d}); VAE conv {r['vae_conv_tflops']:.0f} TFLOP/s, {r['vae_conv_ms_per_pass']:.1f} ms (0.{int(r['vae_conv_tflops']/2500*100):02d})",
 '@@SMALL@@': f"{b[0]:.1f} / {b[1]:.1f} / {b[2]:.1f} / {b[3]:.1f} ms per request; {b1:.1f} ms; {c0:.1f} ms",
 '@@CPU@@': f"{cb['value']:.4f} songs/s on {cb['cores']} threads (DiT extrapolated from 3 forwards, decode from 16 frames); configs[0] in full: {cb['config0_full_run']['seconds']:.2f} s",
 '@@TESTS@@': f"{npass} GPU tests + smoke ({smoke}); 79 CPU tests",
 '@@SPEEDNOTE@@': f"{d['ms_per_step']:.1f} ms on this round's evidence box; the round's boxes gave 456-475 ms for the same library, as in round 5",
 '@@ATTNFRAC@@': f"{100*r['attn_ms_per_pass']/d['ms_per_step']:.0f} %",
}
head=open('/root/repo/tools/r06_docs/head.md').read()
for k,v in rep.items(): head=head.replace(k,v)
head=head.replace('r06_final', TAG)
assert '@@' not in head, re.findall(r'@@\w+@@', head)
old=open(f'{ROOT}/DESIGN.md').read()
# already assembled? then take the body between "## 1. The path" and "## 14." and the appendix from the marker
i1=old.index('## 1. The path')
if '## 14. Round 6' in old:
    body=old[i1:old.index('## 14. Round 6')]
    app=old[old.index('## Appendix A.'):]
else:
    body=old[i1:]
    if not body.endswith('\n'): body+='\n'
    oldhead=old[:i1]
    k=oldhead.index('Status at the end of round 3')
    app='## Appendix A. The status paragraphs and verdict tables of rounds 1-5, as written at the time\n\n(Moved here from the top of the file in round 6; the numbers in them are those rounds\' numbers.)\n\n'+oldhead[k:]
sec14=open('/root/repo/tools/r06_docs/sec14.md').read().replace('r06_final', TAG)
import collections
cost=collections.defaultdict(lambda: collections.defaultdict(list))
for ln in open(f'{ROOT}/profiles/r06_shape_independent_cost.txt'):
    m=re.match(r'b=(\d+)\s+(default|shape-independent)\s+([\d.]+)', ln)
    if m: cost[int(m.group(1))][m.group(2)].append(float(m.group(3)))
names={1:'one song',2:'two',4:'four',8:'eight'}
parts=[]
for bsz in (1,2,4,8):
    dd=sum(cost[bsz]['default'])/len(cost[bsz]['default']); si=sum(cost[bsz]['shape-independent'])/len(cost[bsz]['shape-independent'])
    parts.append(f"{names[bsz]} {si:.1f} vs {dd:.1f} ms (+ {100*(si/dd-1):.1f} %)")
sec14=sec14.replace('@@INVCOST@@', ', '.join(parts))
new=head+body.rstrip('\n')+'\n\n'+sec14.rstrip('\n')+'\n\n'+app
open(f'{ROOT}/DESIGN.md','w').write(new)
print('DESIGN.md', len(new))
