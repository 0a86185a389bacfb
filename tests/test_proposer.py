"""The BERT-MLM candidate proposer (``attack/proposer.py``) against the ``transformers`` library's ``BertForMaskedLM`` --
the class the reference instantiates (``BertForMaskedLM.from_pretrained('bert-base-uncased')``, adv_attack.py:110) -- on
seeded weights: its ``state_dict()`` loads by its own key names and the logits agree; the proposals that come out of
``propose_candidates`` (top-5, threshold 0.3, original word / ``##`` piece / stop-word filters, adv_attack.py:240-258) are
then the same lists.  Host path here; the HIP path (attention on csrc/attn.hip) in the gpu-marked twin."""
import pytest
import torch

from vqattack_amd.attack import text_update
from vqattack_amd.attack.proposer import BertMlmProposer, banned_ids


def _hf_model(heads=2, dim=128, depth=2, vocab=211):
    tf = pytest.importorskip("transformers")
    cfg = tf.BertConfig(vocab_size=vocab, hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                        intermediate_size=4 * dim, max_position_embeddings=48, hidden_dropout_prob=0.0,
                        attention_probs_dropout_prob=0.0)
    torch.manual_seed(3)
    model = tf.BertForMaskedLM(cfg).eval()
    with torch.no_grad():                        # trained-like magnitudes: HF's 0.02 init gives near-uniform logits
        for name, p in model.named_parameters():
            if p.dim() == 2 and "embeddings" not in name:
                p.normal_(0.0, p.shape[1] ** -0.5)
            elif p.dim() == 1 and "LayerNorm" not in name:
                p.normal_(0.0, 0.1)
    return model


def _inputs(vocab, device):
    g = torch.Generator().manual_seed(5)
    ids = torch.zeros(3, 12, dtype=torch.long)
    masks = torch.zeros(3, 12, dtype=torch.long)
    for b, n in enumerate((12, 7, 9)):
        ids[b, 0], ids[b, n - 1] = 101, 102
        ids[b, 1:n - 1] = torch.randint(110, vocab, (n - 2,), generator=g)
        masks[b, :n] = 1
    return ids.to(device), masks.to(device)


def _check(device):
    hf = _hf_model()
    ours = BertMlmProposer.from_hf_state_dict(hf.state_dict(), heads=2).to(device)
    assert len(ours.layers) == 2 and ours.word_embeddings.weight.shape == (211, 128)
    ids, masks = _inputs(211, device)
    with torch.no_grad():
        want = hf(input_ids=ids.cpu(), attention_mask=masks.cpu()).logits
    got = ours(ids, masks).cpu()
    real = masks.cpu().bool()
    err = float((got - want)[real].abs().max()) / float(want[real].abs().max())
    assert err <= 2e-5, err
    # the reference's own call has no padding and no mask (adv_attack.py:240-242)
    with torch.no_grad():
        want1 = hf(input_ids=ids[:1].cpu()).logits
    assert float((ours(ids[:1]).cpu() - want1).abs().max()) <= 2e-5 * float(want1.abs().max())
    att = torch.zeros_like(ids, dtype=torch.bool)
    att[:, 1:4] = True
    ban = banned_ids(["tok{}".format(i) if i % 7 else "##p{}".format(i) for i in range(211)], stop_words={"tok15"})
    assert int(ban.sum()) == 31 + 1
    a = text_update.propose_candidates(got, ids.cpu(), att.cpu(), banned=ban, threshold=0.0)
    b = text_update.propose_candidates(want, ids.cpu(), att.cpu(), banned=ban, threshold=0.0)
    assert a == b and sum(len(c) for per in a for _, c in per) >= 20
    with pytest.raises(Exception):
        BertMlmProposer.from_hf_state_dict({"bert.embeddings.word_embeddings.weight": torch.zeros(5, 8)})


def test_proposer_equals_transformers_bert_for_masked_lm_host():
    _check("cpu")


@pytest.mark.gpu
def test_proposer_equals_transformers_bert_for_masked_lm_hip():
    _check("cuda")
