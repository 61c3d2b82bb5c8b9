import importlib, sys, time, numpy as np, torch
sys.path.insert(0, ".")
PKG="distributed-multi-agent-slam-swarm-robotics-system_amd"
pkg=importlib.import_module(PKG); replay=importlib.import_module(PKG+".replay")
dev=torch.device("cuda",0); side=torch.cuda.Stream(device=dev); torch.cuda.set_stream(side)
B=1<<20
def run64(tag, steps=20):
    stream=replay.multi_bot_stream(None,64,B); d=torch.from_numpy(stream).to(dev); t=torch.arange(B,dtype=torch.float64,device=dev)*0.25
    m=pkg.QuasarMapper(4096,0.05,-102.4,-102.4,max_agent=64,bots_per_graph=2,enable_ekf=True,device=0)
    m.set_stream(side.cuda_stream)
    def step(): m.reset(); m.ingest_device(d.data_ptr(),B,42,0,t.data_ptr(),seq0=0)
    for _ in range(3): step()
    torch.cuda.synchronize(); m.stage_times(reset=True); m.timing_enable(True)
    t0=time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    st={k:round(v[0]/max(v[1],1),3) for k,v in m.stage_times(reset=True).items() if v[1]}
    print(tag, "%.3f ms/step"%(el/steps*1e3), st, flush=True)
    m.close()
mode=sys.argv[1]
if mode=="alone": run64("alone")
elif mode=="after_c1":
    session,_=replay.telemetry_csv_to_packets(); s=replay.cycle_stream(session,B); d=torch.from_numpy(s).to(dev)
    m1=pkg.QuasarMapper(4096,0.05,-102.4,-102.4,max_agent=2,enable_ekf=True,device=0); m1.set_stream(side.cuda_stream)
    for _ in range(3): m1.reset(); m1.ingest_device(d.data_ptr(),B,42,0,0,seq0=0)
    torch.cuda.synchronize()
    run64("after_c1 (c1 mapper alive)")
    m1.close(); run64("after_c1 closed")
elif mode=="after_alloc":
    big=[torch.randint(0,1<<30,(4096*4096,),dtype=torch.int32,device=dev) for _ in range(64)]
    torch.cuda.synchronize(); run64("with 4 GB of torch tensors alive")
    del big; run64("after del (cached by torch)")
    torch.cuda.empty_cache(); run64("after empty_cache")
