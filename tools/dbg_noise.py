import sys, numpy as np, torch
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
from diligentfx_amd import api, synth
from util import blue_noise_tables
import pyref
sobol,tile=blue_noise_tables()
ctx=api.PostFXContext(0,sobol,tile)
for frame in (0,1,17):
    ctx.prepare_resources(frame,64,48)
    z=torch.ones(48,64,device=ctx.device); cam=synth.make_camera(frame,64,48)
    ctx.execute(z,z,torch.zeros(48,64,2,device=ctx.device),cam,cam)
    xy=ctx.get_2d_blue_noise(0).cpu().numpy(); zw=ctx.get_2d_blue_noise(1).cpu().numpy()
    wxy=np.zeros((128,128,2),np.float32); wzw=np.zeros((128,128,2),np.float32)
    pyref.oracle_lib().call('oracle_blue_noise',[sobol.astype(np.float32).reshape(1,256),tile.astype(np.float32).reshape(256,512)],[wxy,wzw],ival=[frame])
    for n,a,b in (('xy',xy,wxy),('zw',zw,wzw)):
        d=np.abs(a-b); print(frame,n,'mismatch',(d>0).sum(),'max',d.max(), 'max*255', (d*255).max())
