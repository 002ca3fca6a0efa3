#!/usr/bin/env python
"""Soak: 250 greedy tokens of the full Mistral-7B-shaped model (32 layers, random init) at 25 % and 50 % effort, the decode loop
with a layer's dependent multiplies as ONE chain launch against the loop with launches of their own: tokens and logits must be
bit-identical (round 4, one box: they are; 4.06 against 3.35 ms per token at 25 %).

    python tools/chain_soak.py
"""
import sys, torch
sys.path.insert(0, '.')
from effort_amd.decode import Decoder, MistralConfig, Model
torch.cuda.set_device(0)
model = Model.random(MistralConfig(), seed=2, keep_cores=False)
a = Decoder(model, maxTokens=260, chain=False)
b = Decoder(model, maxTokens=260, chain=True)
for effort in (0.25, 0.5):
    ia, ta, la = a.run([1, 5, 9], 250, effort=effort, collect_logits=True)
    ib, tb, lb = b.run([1, 5, 9], 250, effort=effort, collect_logits=True)
    print(effort, 'tokens equal', ia == ib, 'logits equal', bool(torch.equal(la, lb)), 'ms/token', round(ta*1e3,3), round(tb*1e3,3), flush=True)
