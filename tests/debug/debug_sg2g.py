import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import stylegan2_oracle as S
from contrad_amd.models.gan.stylegan2.generator import Generator
from contrad_amd import ops
DEV='cuda'
def rel(a,b):
    a,b=a.double().cpu(),b.double().cpu(); return ((a-b).abs().max()/b.abs().max().clamp_min(1e-30)).item()
g=np.load(os.path.join(os.path.dirname(__file__),'..','tests','golden','stylegan2_g.npz'))
shapes=S.g_param_shapes(32,True)
sd=S.fill_kernels(S.det_fill_g(shapes,seed=777),shapes)
G=Generator(size=32,n_mlp=8,small32=True); G.load_state_dict(sd); G=G.to(DEV).train()
z=torch.from_numpy(g['z']); noise=[torch.from_numpy(g['noise%d'%i]) for i in range(G.num_layers)]
with torch.no_grad():
    c=G._prepared()
    lat=S.mapping(sd,z); latd=G._mapping(z.to(DEV),c); print('mapping',rel(latd,lat))
    B=z.shape[0]
    x0=sd['input.const'].repeat(B,1,1,1)
    # oracle conv1 pieces
    mc=G.conv1.conv
    s_o=S.equal_linear(lat, sd['conv1.conv.modulation.weight'], sd['conv1.conv.modulation.bias'], bias_init=1.0)
    s_h=G._style(mc, latd, c); print('style',rel(s_h,s_o))
    mo=S.modulated_conv(sd,'conv1.conv',x0,lat)
    xh=G.input.const.permute(0,2,3,1).expand(B,-1,-1,-1).contiguous()
    wsq=c['wsq'][mc]
    d=ops.conv2d_fwd((s_h*s_h).view(B,1,1,-1), wsq, None, mc.out_channel,1,1,1,0).view(B,-1)
    demod=torch.rsqrt(d+1e-8)
    w=sd['conv1.conv.weight']; scale=1/math.sqrt(512*9)
    weight=scale*w*s_o.view(B,1,512,1,1); demod_o=torch.rsqrt(weight.pow(2).sum([2,3,4])+1e-8)
    print('demod',rel(demod,demod_o))
    xm=ops.nhwc_scale(xh,s_h)
    y=ops.conv2d_fwd(xm,c['packed'][c['conv'][mc]],None,mc.out_channel,3,3,1,1)
    ymod=y*demod.view(B,1,1,-1)
    print('modconv',rel(ymod.permute(0,3,1,2),mo))
    o1=S.styled_layer(sd,'conv1',x0,lat,noise[0])
    h1=G._styled_conv(G.conv1,xh,latd,noise[0].to(DEV),c); print('conv1 layer',rel(h1.permute(0,3,1,2),o1))
    r1=S.to_rgb(sd,'to_rgb1',o1,lat); hr1=G._to_rgb(G.to_rgb1,h1,latd,None,c); print('to_rgb1',rel(hr1,r1))
    o2=S.styled_layer(sd,'layers.0',o1,lat,noise[1],upsample=True)
    h2=G._styled_conv(G.layers[0],h1,latd,noise[1].to(DEV),c); print('layers.0 (up)',rel(h2.permute(0,3,1,2),o2))
    o3=S.styled_layer(sd,'layers.1',o2,lat,noise[2])
    h3=G._styled_conv(G.layers[1],h2,latd,noise[2].to(DEV),c); print('layers.1',rel(h3.permute(0,3,1,2),o3))
    r2=S.to_rgb(sd,'to_rgbs.0',o3,lat,r1); hr2=G._to_rgb(G.to_rgbs[0],h3,latd,hr1,c); print('to_rgbs.0',rel(hr2,r2))
    # full chain
    o, h = o3, h3
    rs, hrs = r2, hr2
    for j in (1, 2):
        o = S.styled_layer(sd,'layers.%d'%(2*j),o,lat,noise[1+2*j],upsample=True)
        h = G._styled_conv(G.layers[2*j],h,latd,noise[1+2*j].to(DEV),c); print('layers.%d'%(2*j),rel(h.permute(0,3,1,2),o))
        o = S.styled_layer(sd,'layers.%d'%(2*j+1),o,lat,noise[2+2*j])
        h = G._styled_conv(G.layers[2*j+1],h,latd,noise[2+2*j].to(DEV),c); print('layers.%d'%(2*j+1),rel(h.permute(0,3,1,2),o))
        rs = S.to_rgb(sd,'to_rgbs.%d'%j,o,lat,rs); hrs = G._to_rgb(G.to_rgbs[j],h,latd,hrs,c); print('to_rgbs.%d'%j,rel(hrs,rs))
    img = G(z.to(DEV), style_mix=0.0, noise=[n.to(DEV) for n in noise])
    print('forward vs chain', rel(img, 0.5*hrs+0.5), 'vs golden', rel(img, torch.from_numpy(g['img_nomix'])), rel(0.5*rs+0.5, torch.from_numpy(g['img_nomix'])))
