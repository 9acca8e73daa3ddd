"""per-queue summary of one rp_step from the kernel traces tools/trace_chain.sh left under gpurun_out/trace_chain_g<G>/ (which stream ran on which
hardware queue, how long its chain was busy, average kernel durations).   python3 tools/trace_groups.py 3 4"""
import csv,glob,os,sys
for G in sys.argv[1:]:
    f=sorted(glob.glob('gpurun_out/trace_chain_g%s/t/**/*kernel_trace.csv'%G,recursive=True),key=os.path.getmtime)[-1]      # the newest run
    rows=list(csv.DictReader(open(f)))
    ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0],r.get('Queue_Id','0')) for r in rows]
    ks.sort()
    starts=[i for i,k in enumerate(ks) if k[2].startswith('k_member') and not k[2].startswith('k_member_id')]
    a,b=starts[8],starts[9]
    step=ks[a:b]; t0=step[0][0]
    print('G',G,' step kernels',len(step),'span %.3f ms'%((max(k[1] for k in step)-t0)/1e6))
    byq={}
    for k in step: byq.setdefault(k[3],[]).append(k)
    for q,l in byq.items():
        names={}
        for k in l: names.setdefault(k[2][:14],[]).append((k[1]-k[0])/1e3)
        print('  queue',q,len(l),'busy %.3f span %.3f first %.1f last %.1f'%(sum(k[1]-k[0] for k in l)/1e6,(l[-1][1]-l[0][0])/1e6,(l[0][0]-t0)/1e3,(l[-1][1]-t0)/1e3),{n:round(sum(v)/len(v),1) for n,v in names.items()})
