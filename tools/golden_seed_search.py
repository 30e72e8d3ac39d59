import sys; sys.dont_write_bytecode=True
sys.path.insert(0,'/root/repo/tests/golden'); sys.path.insert(0,'/root/repo')
import numpy as np, torch
import _reference_import as R
R.install()
import procedural as P
import torch.nn.functional as F
from multivae.models.nn.cub import CUB_Resnet_Decoder, CUB_Resnet_Encoder
from multivae.models.nn.mmnist import DecoderResnetMMNIST, EncoderResnetMMNIST
torch.set_num_threads(8)
t=lambda a: torch.from_numpy(np.ascontiguousarray(a))
rec=[]
orig=F.leaky_relu
def patched(x,*a,**k):
    ax=x.detach().abs()
    rec.append(float((ax/ax.max()).min()))
    return orig(x,*a,**k)
F.leaky_relu=patched
which=sys.argv[1]; s0=int(sys.argv[2]); n=int(sys.argv[3])
best=[]
for seed in range(s0,s0+n):
    rec.clear()
    with torch.no_grad():
        if which=='mmnist':
            B,K,pd_,sd_=3,2,4,6; L=pd_+sd_
            enc, dec = EncoderResnetMMNIST(pd_, sd_), DecoderResnetMMNIST(L)
            enc.load_state_dict({k: t(v) for k, v in P.make_state_dict(P.mmnist_resnet_encoder_shapes(pd_, sd_), seed).items()})
            dec.load_state_dict({k: t(v) for k, v in P.make_state_dict(P.mmnist_resnet_decoder_shapes(L), seed+1).items()})
            enc(t(P.uniform((B, 3, 28, 28), seed + 2))); dec(t(P.uniform((K, B, L), seed + 3, -1.0, 1.0)))
        else:
            L=12;B=2
            enc, dec = CUB_Resnet_Encoder(L), CUB_Resnet_Decoder(L)
            enc.load_state_dict({k: t(v) for k, v in P.make_state_dict(P.cub_resnet_encoder_shapes(L), seed).items()})
            dec.load_state_dict({k: t(v) for k, v in P.make_state_dict(P.cub_resnet_decoder_shapes(L), seed+1).items()})
            enc(t(P.uniform((B, 3, 64, 64), seed + 2))); dec(t(P.uniform((B, L), seed + 3, -1.0, 1.0)))
    m=min(rec); best.append((m,seed)); print(seed, f"{m:.3e}", flush=True)
best.sort(reverse=True); print("BEST", best[:5])
