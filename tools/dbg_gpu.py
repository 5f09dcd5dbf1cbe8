import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests'); sys.path.insert(0, ROOT+'/tools')
import numpy as np, torch
from fastp_amd import abi, engine
import oraclelib, synth_torch
dev = torch.device('cuda',0)
p = abi.default_params(True,150); p.cut_right=1
n = int(sys.argv[1]) if len(sys.argv)>1 else 200000
t=time.time(); d = synth_torch.synth_pairs_torch(n, L=150, seed=1000, device=dev); torch.cuda.synchronize(); print('gen', time.time()-t)
s1,q1,l1 = synth_torch.pack_torch(d['seq1'],d['qual1'],d['len1'],150); s2,q2,l2 = synth_torch.pack_torch(d['seq2'],d['qual2'],d['len2'],150)
g = engine.GpuEngine(p)
r1 = torch.zeros(n*12, dtype=torch.uint8, device=dev); r2=torch.zeros(n*12,dtype=torch.uint8,device=dev); pr=torch.zeros(n*8,dtype=torch.uint8,device=dev); nc=torch.zeros(1,dtype=torch.int32,device=dev)
b=abi.Batch(); b.n=n; b.flags=1; b.seq1,b.qual1,b.len1=s1.data_ptr(),q1.data_ptr(),l1.data_ptr(); b.seq2,b.qual2,b.len2=s2.data_ptr(),q2.data_ptr(),l2.data_ptr()
res=abi.Results(); res.r1,res.r2,res.pair=r1.data_ptr(),r2.data_ptr(),pr.data_ptr(); res.corrections=None; res.corrections_capacity=0; res.n_corrections=nc.data_ptr()
t=time.time(); g.submit_device(b,res); g.synchronize(); torch.cuda.synchronize(); print('submit', time.time()-t, 'kernel', g.kernel_time())
R1=r1.cpu().numpy().view(abi.READ_RESULT_DTYPE); R2=r2.cpu().numpy().view(abi.READ_RESULT_DTYPE); PR=pr.cpu().numpy().view(abi.PAIR_RESULT_DTYPE)
cg=g.counters(); lay=g.layout
print('gpu filter', cg[lay.filter_stats:lay.filter_stats+32][[0,12,16,17,20,24,28]], 'records pass', int(((R1['code']==0)&(R2['code']==0)).sum()))
m = min(n, 100000)
pad=lambda a: np.pad(a[:m].cpu().numpy(),((0,0),(0,2)))
o=oraclelib.Oracle(p)
ro=o.process(pad(d['seq1']),pad(d['qual1']),d['len1'][:m].cpu().numpy(),pad(d['seq2']),pad(d['qual2']),d['len2'][:m].cpu().numpy())
co=o.counters()
print('oracle filter (first m)', co[lay.filter_stats:lay.filter_stats+32][[0,12,16,17,20,24,28]])
for k,(a,bb) in enumerate(((ro[0],R1[:m]),(ro[1],R2[:m]),(ro[2],PR[:m]))):
    bad=np.nonzero(a!=bb)[0]; print('records', k, 'mismatch', len(bad), bad[:5], a[bad[:3]], bb[bad[:3]])
if n==m:
    bad=np.nonzero(co!=cg)[0]; print('counter mismatches', len(bad), bad[:10], co[bad[:10]], cg[bad[:10]])
# same data through the host path in one go
g2=engine.GpuEngine(p)
rh=g2.submit_packed(s1.cpu().numpy(),q1.cpu().numpy(),l1.cpu().numpy().view(np.uint16),s2.cpu().numpy(),q2.cpu().numpy(),l2.cpu().numpy().view(np.uint16))
c2=g2.counters()
print('host path equal records', rh[0].tobytes()==R1.tobytes(), rh[1].tobytes()==R2.tobytes(), 'counters equal', np.array_equal(c2,cg))

# two passes of a 2M batch: print all filter bins after each
n2 = 2*1024*1024
d = synth_torch.synth_pairs_torch(n2, L=150, seed=1000, device=dev)
s1,q1,l1 = synth_torch.pack_torch(d['seq1'],d['qual1'],d['len1'],150); s2,q2,l2 = synth_torch.pack_torch(d['seq2'],d['qual2'],d['len2'],150)
del d
g3 = engine.GpuEngine(p); lay=g3.layout
r1 = torch.zeros(n2*12, dtype=torch.uint8, device=dev); r2=torch.zeros(n2*12,dtype=torch.uint8,device=dev); pr=torch.zeros(n2*8,dtype=torch.uint8,device=dev)
b=abi.Batch(); b.n=n2; b.flags=1; b.seq1,b.qual1,b.len1=s1.data_ptr(),q1.data_ptr(),l1.data_ptr(); b.seq2,b.qual2,b.len2=s2.data_ptr(),q2.data_ptr(),l2.data_ptr()
res=abi.Results(); res.r1,res.r2,res.pair=r1.data_ptr(),r2.data_ptr(),pr.data_ptr(); res.corrections=None; res.corrections_capacity=0; res.n_corrections=nc.data_ptr()
prev = g3.counters()
for it in range(4):
    g3.submit_device(b,res); g3.synchronize(); torch.cuda.synchronize()
    c = g3.counters(); dlt = c - prev; prev = c
    R1=r1.cpu().numpy().view(abi.READ_RESULT_DTYPE); R2=r2.cpu().numpy().view(abi.READ_RESULT_DTYPE)
    fs = dlt[lay.filter_stats:lay.filter_stats+32]
    print('pass', it, 'filter nonzero', {int(i):int(v) for i,v in enumerate(fs) if v}, 'sum', int(fs.sum()), 'records pass', int(((R1['code']==0)&(R2['code']==0)).sum()),
          'dup', int(dlt[lay.dup_count]), int(dlt[lay.dup_total]), 'pre1 reads', int(dlt[lay.stats[0]+lay.st_reads]), 'post1 reads', int(dlt[lay.stats[1]+lay.st_reads]), 'kernel', g3.kernel_time())
