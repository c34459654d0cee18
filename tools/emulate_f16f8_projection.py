"""Emulation (torch CPU) of the split path's input projection with W_ih ~ fp16(W) + e4m3((W - fp16(W)) 2^15) 2^-15 and the residual product on e4m3(x),
against the reference forward on the north-star golden: the evidence behind lstm_sp.hip's x_convert / xproj_sp_kernel arithmetic (round 3).
Modes: exact | f16+8 (what the kernels do) | f16only | bf16+8.  Run from the repo root: python tools/emulate_f16f8_projection.py"""
import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import torch.nn.functional as F
from torch import _VF
torch.set_num_threads(8)
d=np.load('/root/repo/tests/golden/vae_fwd_ns_h128.npz')
P={k[2:]:torch.tensor(d[k]) for k in d.files if k.startswith('p.')}
x=torch.tensor(d['x']); eps=torch.tensor(d['eps'])
H=128
def e4m3(t): return t.to(torch.float8_e4m3fn).float()
def proj(xin, W, b, mode):
    if mode=='exact': return xin@W.T + b
    W16=W.half().float()
    dW=(W-W16)*2**15
    dW8=e4m3(dW.clamp(-224,224))
    x8=e4m3(xin.clamp(-224,224))
    if mode=='f16only': return xin@W16.T + b
    if mode=='bf16+8':
        Wb=W.bfloat16().float(); dWb=e4m3(((W-Wb)*2**12).clamp(-224,224))
        return xin@Wb.T + (x8@dWb.T)*2**-12 + b
    return (x8@dW8.T)*2**-15 + xin@W16.T + b
def lstm_layer(xin, pre, l, mode, const=False):
    W_ih=P[f'{pre}.weight_ih_l{l}']; W_hh=P[f'{pre}.weight_hh_l{l}']; b=P[f'{pre}.bias_ih_l{l}']+P[f'{pre}.bias_hh_l{l}']
    B=xin.size(0)
    if mode is None:
        flat=[W_ih,W_hh,P[f'{pre}.bias_ih_l{l}'],P[f'{pre}.bias_hh_l{l}']]
        inp=xin
    else:
        pr=proj(xin,W_ih,b,mode)
        flat=[torch.eye(4*H),W_hh,torch.zeros(4*H),torch.zeros(4*H)]
        inp=pr
    h0=xin.new_zeros(1,B,H)
    out,hn,_=_VF.lstm(inp,(h0,h0.clone()),flat,True,1,0.0,False,False,True)
    return out,hn
def fwd(mode):
    o0,_=lstm_layer(x,'encoder_lstm',0,None)
    o1,hn=lstm_layer(o0,'encoder_lstm',1,mode.replace('-dec0exact',''))
    h=F.layer_norm(hn[-1],(H,),P['layer_norm.weight'],P['layer_norm.bias'],1e-5)
    mu=F.linear(h,P['fc_mu.weight'],P['fc_mu.bias']); lv=F.linear(h,P['fc_logvar.weight'],P['fc_logvar.bias'])
    z=mu+eps*torch.exp(0.5*lv)
    h0=torch.tanh(F.linear(z,P['fc_latent_to_hidden.weight'],P['fc_latent_to_hidden.bias']))
    xin=h0.unsqueeze(1).repeat(1,x.size(1),1)
    d0,_=lstm_layer(xin,'decoder_lstm',0,'exact' if mode.endswith('-dec0exact') else mode)
    d1,_=lstm_layer(d0,'decoder_lstm',1,mode.replace('-dec0exact',''))
    rec=F.linear(d1,P['output_layer.weight'],P['output_layer.bias'])
    return rec,mu,lv
with torch.no_grad():
    for mode in ['exact','f16+8','f16+8-dec0exact','f16only','bf16+8']:
        rec,mu,lv=fwd(mode)
        sc=((rec-x)**2).mean(dim=(1,2)).numpy()
        print(f"{mode:8s} recon {np.abs(rec.numpy()-d['recon']).max():.2e} mu {np.abs(mu.numpy()-d['mu']).max():.2e} lv {np.abs(lv.numpy()-d['logvar']).max():.2e} score {np.abs(sc-d['score']).max():.2e}")
