"""CPU: the batched pretrain input preparation (visitron_amd/data.py) against the per-item restatement of
data_loader_pretrain.py (oracle/data.py) on the same random draws, plus the distribution of its own draws."""
import torch

from oracle import data as odata
from visitron_amd import data as vdata
from visitron_amd.synth import viewpoint_loc_embedding


def test_location_embedding_table_matches_reference_construction():
    t = vdata.loc_embedding_table()
    assert t.shape == (36, 36, 128)
    for v in (0, 7, 35):
        assert torch.allclose(t[v], torch.from_numpy(odata.STATIC[v]), atol=1e-6)
        assert torch.allclose(t[v], viewpoint_loc_embedding(v), atol=1e-6)


def test_mask_tokens_and_assembly_match_per_item_restatement():
    check_against_per_item_restatement("cpu")


def check_against_per_item_restatement(device):
    """visitron_amd.data on `device` against oracle/data.py item by item, on shared random draws.  Integer outputs
    (ids, labels, attention mask, token labels, next action) must be EQUAL; the float outputs are gathers of the inputs
    / of the location table, so they must be equal too (the table to fp32 rounding of sin / cos)."""
    dev = torch.device(device)
    g = torch.Generator().manual_seed(5)
    B, T, V, R_in, D, R = 6, 24, 997, 30, 11, 20
    special_ids, pad_id, mask_id = {0, 101, 102, 103}, 0, 103
    ids = torch.randint(200, V, (B, T), generator=g)
    ids[:, 0] = 101
    for b in range(B):
        ids[b, T - 1 - b:] = 0                                  # ragged padding
        ids[b, T - 2 - b] = 102
    token_classes = torch.full((B, T), -1, dtype=torch.long)
    token_classes[:, 3] = torch.randint(0, 40, (B,), generator=g)
    special = torch.zeros(B, T, dtype=torch.bool)
    for s in special_ids:
        special |= ids == s
    draws = (torch.rand(B, T, generator=g), torch.rand(B, T, generator=g), torch.rand(B, T, generator=g),
             torch.randint(V, (B, T), generator=g))
    for tc in (None, token_classes):
        d = lambda t: t.to(dev)
        got_in, got_lab, got_att = vdata.mask_tokens(d(ids), d(special), pad_id, mask_id, V, 0.15,
                                                     token_classes=None if tc is None else d(tc),
                                                     draws=tuple(d(x) for x in draws))
        counts = torch.tensor([30, 20, 25, 5, 0, 19])            # > R, == R, between, short, empty, R - 1
        feats = torch.rand(B, R_in, D, generator=g)
        view_ids = torch.randint(0, 36, (B, R_in), generator=g)
        cur = torch.randint(0, 36, (B,), generator=g)
        nxt = torch.randint(0, 36, (B,), generator=g)
        batch = vdata.assemble_batch(got_in, got_lab, got_att, d(feats), d(counts), d(view_ids), d(cur), d(nxt), R,
                                     token_classes=None if tc is None else d(tc))
        assert all(v is None or v.device.type == dev.type for v in batch.values())
        batch = {k: (None if v is None else v.cpu()) for k, v in batch.items()}
        got_in, got_lab, got_att = got_in.cpu(), got_lab.cpu(), got_att.cpu()
        for b in range(B):
            w_in, w_lab, w_att = odata.mask_tokens_item(ids[b], special_ids, pad_id, mask_id, 0.15, None if tc is None else tc[b],
                                                        draws[0][b], draws[1][b], draws[2][b], draws[3][b])
            assert torch.equal(got_in[b], w_in) and torch.equal(got_lab[b], w_lab) and torch.equal(got_att[b], w_att)
            n = int(counts[b])
            want = odata.preprocess_item_tail(w_in, w_lab, w_att, feats[b, :n].numpy(), view_ids[b, :n].tolist(), int(cur[b]),
                                              int(nxt[b]), R, token_classes=None if tc is None else tc[b])
            assert torch.equal(batch["labels"][b], want["labels"])
            assert torch.equal(batch["attention_mask"][b], want["attention_mask"].long())
            assert torch.equal(batch["img_feats"][b], want["img_feats"])
            assert torch.allclose(batch["img_location_embeddings"][b], want["img_location_embeddings"], atol=1e-6, rtol=0)
            if tc is not None:
                assert torch.equal(batch["token_labels"][b], want["token_labels"])
            assert int(batch["next_action"][b]) == want["next_action"]
        assert batch["img_feats"].shape == (B, R, D) and batch["attention_mask"].shape == (B, T + R)


def test_mask_tokens_own_draws_follow_the_80_10_10_rule():
    g = torch.Generator().manual_seed(11)
    B, T, V = 64, 512, 30522
    ids = torch.randint(1000, V, (B, T), generator=g)
    special = torch.zeros(B, T, dtype=torch.bool)
    special[:, 0] = True
    out, labels, att = vdata.mask_tokens(ids, special, 0, 103, V, 0.15, generator=g)
    sel = labels != -1
    n = int(sel.sum())
    assert abs(n / (B * (T - 1)) - 0.15) < 0.01 and not bool(sel[:, 0].any()) and bool(att.all())
    assert torch.equal(labels[sel], ids[sel])
    frac_mask = float((out[sel] == 103).float().mean())
    frac_same = float((out[sel] == ids[sel]).float().mean())
    assert abs(frac_mask - 0.8) < 0.02 and abs(frac_same - 0.1) < 0.02
    assert torch.equal(out[~sel], ids[~sel])
