"""The one-row top block of the loss path as the TAIL of the forward launch of the block below it and as the HEAD of that
block's backward launch (fused_layer.h TAILP / HEADP; plan option separate_top = 0, the default) against the same
block as kernels of its own (separate_top = 1): the arithmetic is the same instruction for instruction, only the x / dX
tiles stay in LDS instead of a round trip through global memory -- loss and every gradient must be bit-identical (the
item table's up to the order of its atomic row additions), with
dropout on (same Philox stream), in fp32 and in bf16 storage, for 2 and 3 blocks."""
import argparse

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _ns(**kw):
    a = argparse.Namespace(item_size=211, hidden_size=64, max_seq_length=50, batch_size=24, hidden_dropout_prob=0.3,
                           attention_probs_dropout_prob=0.2, num_hidden_layers=2, num_attention_heads=2,
                           hidden_act="gelu", initializer_range=0.05, c=5, alpha=0.7, seed=9)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _run(ns, separate):
    from bsarec_amd import BSARecModel
    ns.plan_options = {"separate_top": separate}
    torch.manual_seed(4)
    m = BSARecModel(ns).cuda()
    m.train()
    m.set_seed(77)
    g = torch.Generator(device="cpu").manual_seed(1)
    ids = torch.randint(1, ns.item_size, (ns.batch_size, ns.max_seq_length), generator=g)
    ids[:, :7] = 0
    ids[3, :] = 0
    ids[3, -1] = 5
    ans = torch.randint(1, ns.item_size, (ns.batch_size,), generator=g)
    out = []
    for _ in range(2):                      # two steps: the dropout step counter advances the same way
        m.zero_grad()
        loss = m.calculate_loss(ids.cuda(), ans.cuda())
        loss.backward()
        out.append((float(loss.detach()), {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}))
    return out


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("kw", [dict(), dict(num_hidden_layers=3, num_attention_heads=4, max_seq_length=37, c=9),
                                dict(num_attention_heads=1, max_seq_length=64, batch_size=5)],
                         ids=["N2_h2_L50", "N3_h4_L37", "N2_h1_L64"])
def test_top_block_inside_the_lower_blocks_launches_is_bit_identical(kw, storage):
    kw = dict(kw)
    if storage == "bf16":
        kw["storage"] = "bf16"
    a = _run(_ns(**kw), 1)
    b = _run(_ns(**kw), 0)
    for (la, ga), (lb, gb) in zip(a, b):
        assert la == lb
        assert set(ga) == set(gb)
        for k in ga:
            if k == "item_embeddings.weight":      # lookup rows are scattered with float atomics: order-dependent last bits
                np.testing.assert_allclose(ga[k], gb[k], rtol=0, atol=2e-7, err_msg=k)
            else:
                np.testing.assert_array_equal(ga[k], gb[k], err_msg=k)
