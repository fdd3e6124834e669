import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import PolicyNet, ValueNet
from marl_amd.utils import ppo_device
th.manual_seed(0)
dev = "cuda"
N, A, R = 10, 11, 2500
pol = PolicyNet(9 + N, A, 64).to(dev)
val = ValueNet(9, 64).to(dev)
with th.no_grad():
    for p in pol.parameters():
        p.mul_(2.0)
obs = th.rand(R, 9, device=dev)
obs[:, 3] *= 5
act = th.randint(0, A, (R,), device=dev)
adv = th.randn(R, device=dev)
ret = th.randn(R, device=dev) * 3
eye = th.eye(N, device=dev)
x_all = th.cat([obs, eye[th.arange(R, device=dev) % N]], -1)
with th.no_grad():
    lp_old = th.distributions.Categorical(logits=pol(x_all)).log_prob(act) + 0.05 * th.randn(R, device=dev)
idx = th.randperm(R, device=dev)[:640]
clip, cH, cV = 0.2, 8e-4, 0.2
# autograd
d = th.distributions.Categorical(logits=pol(x_all[idx]))
lp = d.log_prob(act[idx])
ratio = th.exp(lp - lp_old[idx])
pl = -th.min(ratio * adv[idx], th.clamp(ratio, 1 - clip, 1 + clip) * adv[idx]).mean()
ent = d.entropy().mean()
vl = ((val(obs[idx]) - ret[idx].unsqueeze(-1)) ** 2).mean()
(pl - cH * ent + cV * vl).backward()
ga = th.cat([p.grad.reshape(-1) for p in pol.parameters()]).clone()
gv = th.cat([p.grad.reshape(-1) for p in val.parameters()]).clone()
fp, fv = ppo_device.FlatGrads(pol), ppo_device.FlatGrads(val)
losses = th.zeros(2, device=dev)
_, e2 = ppo_device.actor_pass(pol, obs, None, act, None, N, old_logp=lp_old, adv=adv, eps_clip=clip, entropy_coef=cH,
                              inv_denom=1.0 / 640, grads=fp.flat, loss=losses[0:1], want_entropy=True, row_index=idx, logits=True)
ppo_device.critic_pass(val, obs, None, 0, target=ret, inv_denom=cV / 640, grads=fv.flat, loss=losses[1:2], row_index=idx)
th.cuda.synchronize()
print("loss", float(pl - cH * ent), float(losses[0]), float(cV * vl), float(losses[1]), float(ent), float(e2.mean()))
for name, a, b in (("policy", ga, fp.flat), ("value", gv, fv.flat)):
    print(name, "max|g|", float(a.abs().max()), "max diff", float((a - b).abs().max()), "rel", float((a - b).norm() / a.norm()))
off = 0
for n, p in pol.named_parameters():
    k = p.numel()
    print(n, float((ga[off:off + k] - fp.flat[off:off + k]).abs().max()), float(ga[off:off + k].abs().max()))
    off += k
