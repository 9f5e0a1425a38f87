#!/usr/bin/env python3
"""Where does the bf16 output error of the head come from?  CPU emulation on the ORACLE (test infrastructure — this script
lives under tests/ for that reason) of the rounding sites of the bf16 path at a reference golden case: GEMM operands
(activations / weights) rounded to bf16, stored intermediates (q, k, v, P, O, MLP hidden) rounded to bf16, fp32 accumulation,
fp32 residual stream, fp32 object-query stream — each class switched on alone — and, second part, the weights rounded one
parameter group at a time.  Output of `python tests/bf16_rounding_sites.py cfg2_b1_video` (8 cores, ~2 min) is committed as
profiles/round2_bf16_output_error.md."""
import sys, math, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import svol_oracle as O
from tests.helpers import head_case
torch.set_num_threads(8)
rb=lambda t: t.to(torch.bfloat16).to(torch.float32)
MODE={'w':True,'x':True,'inner':True,'n':None}
orig_linear=O.linear; O_mha=O.mha
def lin(x,w,b=None):
    isq = (x.dim()==3 and x.shape[1]==MODE['n']) or x.dim()==4
    if isq: return orig_linear(x,w,b)
    xx = rb(x) if MODE['x'] else x
    ww = rb(w) if MODE['w'] else w
    y=xx@ww.transpose(-1,-2)
    return y if b is None else y+b
def mha(q_in,k_in,v_in,in_w,in_b,out_w,out_b,h,key_padding_mask=None,need_output=True):
    B,Lq,d=q_in.shape; Lk=k_in.shape[1]; dh=d//h
    if not need_output or (Lq==MODE['n'] and Lk==MODE['n']):
        return O_mha(q_in,k_in,v_in,in_w,in_b,out_w,out_b,h,key_padding_mask,need_output)
    r = rb if MODE['inner'] else (lambda t:t)
    q=lin(q_in,in_w[:d],in_b[:d])
    k=r(lin(k_in,in_w[d:2*d],in_b[d:2*d]))
    q=r(q*(1.0/math.sqrt(dh)))
    q=q.view(B,Lq,h,dh).transpose(1,2); k=k.view(B,Lk,h,dh).transpose(1,2)
    v=r(lin(v_in,in_w[2*d:],in_b[2*d:])).view(B,Lk,h,dh).transpose(1,2)
    outs=[]
    for hh in range(h):
        s=q[:,hh]@k[:,hh].transpose(-1,-2)
        if key_padding_mask is not None: s=s.masked_fill(key_padding_mask[:,None,:],float('-inf'))
        m=s.max(-1,keepdim=True).values
        e=torch.exp(s-m); l=e.sum(-1,keepdim=True)
        outs.append((r(e)@v[:,hh])/l)
    o=r(torch.stack(outs,1).transpose(1,2).reshape(B,Lq,d))
    return lin(o,out_w,out_b), None
def mlp_block(x,sd,p):
    qside = x.shape[1]==MODE['n']
    hdn=O.gelu_erf(lin(x,sd[p+'.fc1.weight'],sd[p+'.fc1.bias']))
    if not qside and MODE['inner']: hdn=rb(hdn)
    return lin(hdn,sd[p+'.fc2.weight'],sd[p+'.fc2.bias'])
name=sys.argv[1] if len(sys.argv)>1 else 'cfg2_b1_video'
z,meta,args,sd,inp,tg=head_case(name)
MODE['n']=args.num_queries
with torch.no_grad():
    ref_l=torch.from_numpy(z['pred_logits'])
    O.linear,O.mha,O.mlp_block=lin,mha,mlp_block
    for tag,(w,x,inner) in {'all':(1,1,1),'weights only':(1,0,0),'activations only (operands)':(0,1,0),'stored intermediates only (q,k,v,P,O,hid)':(0,0,1),'none':(0,0,0)}.items():
        MODE.update(w=bool(w),x=bool(x),inner=bool(inner))
        out=O.svanet_forward(sd,args,inp['src_sketch'],inp['src_sketch_mask'],inp['src_video'],inp['src_video_mask'])
        e=(out['pred_logits']-ref_l).abs()
        print(f'{name} {tag}: max {e.max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e}',flush=True)

# ---- second part: weights rounded one parameter group at a time (everything else exact) ----
O.linear, O.mha = orig_linear, O_mha
import importlib
importlib.reload(O)
rb = lambda t: t.to(torch.bfloat16).to(torch.float32)

groups={'input_proj':lambda k:'input_' in k,
 'sa_qk':lambda k:'content_self_attn.in_proj_weight' in k,   # includes v rows; refined below
 'sa_out':lambda k:'content_self_attn.out_proj.weight' in k,
 'mlp1_fc1':lambda k:'mlp1.fc1.weight' in k,
 'mlp1_fc2':lambda k:'mlp1.fc2.weight' in k,
 'ca_kv':lambda k:'content_token_cross_attn.in_proj_weight' in k,
 'gate':lambda k:'sketch_video_cross_attn.in_proj_weight' in k,
 'ALL video-side':lambda k: any(t in k for t in ('input_','content_self_attn','mlp1.','content_token_cross_attn.in_proj','sketch_video_cross_attn.in_proj')) and k.endswith('weight') and 'norm' not in k and 'LayerNorm' not in k}
with torch.no_grad():
    for g,f in groups.items():
        sd2={k:(rb(v) if (f(k) and v.dim()==2) else v) for k,v in sd.items()}
        if g=='ca_kv':
            d=256
            for k in sd2:
                if 'content_token_cross_attn.in_proj_weight' in k:
                    w=sd[k].clone(); w[d:]=rb(w[d:]); sd2[k]=w   # only k,v rows (q row is fp32 in the product)
        out=O.svanet_forward(sd2,args,inp['src_sketch'],inp['src_sketch_mask'],inp['src_video'],inp['src_video_mask'])
        e=(out['pred_logits']-ref_l).abs()
        print(f'{g}: max {e.max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e}',flush=True)
