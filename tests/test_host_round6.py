"""CPU: round-6 host logic -- visitron_amd.parallel's scatter (the reference's multi-gpu-dp mode, pretrain.py:93-94)."""
import pytest
import torch


def test_data_parallel_scatter_chunks_like_torch_chunk():
    from visitron_amd.parallel import _scatter

    cpu = [torch.device("cpu")] * 3
    x = torch.arange(10).view(5, 2)
    parts, n = _scatter(x, 3, cpu, 0)
    assert n == 3 and [p.shape[0] for p in parts] == [2, 2, 1] and torch.equal(torch.cat(parts), x)
    parts, n = _scatter(torch.arange(2), 3, cpu, 0)             # fewer rows than replicas: fewer chunks, as torch.nn.DataParallel
    assert n == 2 and len(parts) == 2
    kw = dict(input_ids=torch.zeros(4, 7), labels=torch.ones(4, 7), text_only=False, nested=(torch.zeros(4), None))
    parts, n = _scatter(kw, 2, cpu[:2], 0)
    assert n == 2 and all(set(p) == set(kw) for p in parts)
    assert parts[0]["input_ids"].shape == (2, 7) and parts[1]["labels"].shape == (2, 7)
    assert parts[0]["text_only"] is False and parts[1]["nested"][1] is None and parts[1]["nested"][0].shape == (2,)
    parts, n = _scatter((), 2, cpu[:2], 0)                      # nothing to split: handed to every replica
    assert n == 2 and parts == [(), ()]
    scalar, n = _scatter(torch.tensor(3.0), 2, cpu[:2], 0)      # 0-d tensors go to every replica whole
    assert n == 2 and float(scalar[1]) == 3.0


def test_data_parallel_needs_a_gpu():
    from visitron_amd.parallel import DataParallel

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="needs a GPU"):
        DataParallel(torch.nn.Linear(2, 2))
