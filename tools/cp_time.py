"""Development tool: where the time of wwhip.evaluate.clip_posteriors goes (2048 synthetic clips)."""
import os,sys,time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0]=[ROOT, os.path.join(ROOT,"wakeword-detection_amd")]
import numpy as np, torch
torch.cuda.set_device(0)
from wwhip.evaluate import synth_testset, clip_posteriors
from wwhip.models import engine_for
from wwhip.engine import frontend_params
eng=engine_for(os.path.join(ROOT,"wakeword-detection_amd/assets/tf_lite_models/CRNN_softmax"),0)
clips,labels=synth_testset(2048)
for _ in range(3):
    t0=time.perf_counter(); clip_posteriors(eng,clips); print("clip_posteriors s", round(time.perf_counter()-t0,4))
# segments
fp=frontend_params(); T=eng.window; PAD=8000; hop=2; n=len(clips)
t=[time.perf_counter()]
lens=np.array([len(c) for c in clips],np.int64); soffs=np.concatenate(([0],np.cumsum(lens+2*PAD))); need=int(soffs[-1])+16
pin=torch.empty(need,dtype=torch.int16,pin_memory=True); t.append(time.perf_counter())
pcm=pin.numpy()
for i,c in enumerate(clips):
    a=int(soffs[i]); pcm[a:a+PAD]=0; pcm[a+PAD:a+PAD+len(c)]=c; pcm[a+PAD+len(c):a+2*PAD+len(c)]=0
t.append(time.perf_counter())
d=pin.cuda(non_blocking=True); torch.cuda.synchronize(); t.append(time.perf_counter())
nf_pad=(lens+2*PAD-512)//160+1; foffs=np.concatenate(([0],np.cumsum(nf_pad))); total_f=int(foffs[-1])
d_so,d_fo=torch.from_numpy(soffs).cuda(),torch.from_numpy(foffs).cuda()
d_mel=torch.empty((total_f,40),dtype=torch.float32,device="cuda"); torch.cuda.synchronize(); t.append(time.perf_counter())
eng.logmel_dev(d.data_ptr(),d_so.data_ptr(),d_fo.data_ptr(),n,total_f,int(nf_pad.max()),d_mel.data_ptr(),fp); eng.ctx.synchronize(); t.append(time.perf_counter())
nw=np.where(nf_pad>=T,(nf_pad-T)//hop+1,0); woffs=np.concatenate(([0],np.cumsum(nw)))
slide_row=np.repeat(foffs[:-1]-hop*woffs[:-1],nw)+hop*np.arange(int(woffs[-1]),dtype=np.int64)
d_row=torch.from_numpy(slide_row).cuda(); d_valid=torch.full((len(slide_row),),T,dtype=torch.int32,device="cuda")
d_out=torch.empty((len(slide_row),eng.n_out),dtype=torch.float32,device="cuda"); torch.cuda.synchronize(); t.append(time.perf_counter())
eng.forward_windows_dev(d_mel.data_ptr(),total_f,d_row.data_ptr(),d_valid.data_ptr(),len(slide_row),d_out.data_ptr()); eng.ctx.synchronize(); t.append(time.perf_counter())
post=d_out.cpu().numpy(); t.append(time.perf_counter())
names=["pin alloc","host fill","H2D pcm","small allocs","logmel","window arrays","forward windows","D2H"]
print({k:round((b-a)*1e3,2) for k,a,b in zip(names,t[:-1],t[1:])}, "MB", need*2/1e6, "windows", len(slide_row))
