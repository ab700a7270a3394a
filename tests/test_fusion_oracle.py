"""CPU: the fine-grained fused-backbone restatement (oracle/fusion_ref.py) against the fixtures the REFERENCE's
own FusionSwinTransformer.forward produced (oracle/gen_fusion_golden.py): four stage maps, text states, every parameter's
gradient norm and sampled gradients, for two image sizes with different padding patterns."""
import numpy as np
import pytest
import torch

from oracle import cases, detgen, fusion_ref
from oracle.gen_fusion_golden import FG_CASES, fg_inputs, projections

RT, AT = 2e-4, 2e-5


@pytest.mark.parametrize("name", list(FG_CASES))
def test_fusion_ref_matches_reference_golden(name, golden):
    gold = golden(name)
    torch.set_num_threads(8)
    m = detgen.fill_(fusion_ref.FusionRef().eval())
    assert sorted(n for n, _ in m.named_parameters()) == gold["param_names"].tolist()      # checkpoint-key parity with the reference
    img, ids, am = fg_inputs(name)
    outs, lang = m(ids, am, img)
    for i, o in enumerate(outs):
        cases.check_summary(f"stage{i + 2}", o, gold, RT, AT)
    cases.check_summary("hidden", lang["hidden"], gold, RT, AT)
    cases.check_summary("aggregate", lang["aggregate"], gold, RT, AT)
    projections(name, outs, lang["hidden"]).backward()
    assert gold["unused_params"].size == 0
    for n, p in m.named_parameters():
        gn = float(gold[f"gradnorm/{n}"])
        assert abs(p.grad.double().norm().item() - gn) <= 2e-3 * gn + 1e-7, n
    P = dict(m.named_parameters())
    for key in gold:
        if key.startswith("grad/") and key.endswith("/sub"):
            n = key[len("grad/"):-len("/sub")]
            cases.check_summary("grad/" + n, P[n].grad, gold, 2e-3, 1e-6)
