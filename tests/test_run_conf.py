"""run.py host logic that needs no GPU: the .conf overrides (run.py:338-347 of the reference) and the scalar log."""
import configparser
import json
import os

import pytest


def _args(**kw):
    import run
    args = run.build_parser().parse_args([])
    for k, v in kw.items():
        setattr(args, k, v)
    return run, args


@pytest.mark.parametrize("dataset,model,expect", [
    ("drmm_tks", "attncut", dict(lr=3e-5, batch_size=63, dropout=0.1, weight_decay=0.0014756345581373493)),
    ("drmm_tks", "moecut", dict(lr=3e-5, dropout=0.0, rerank_weight=0.2, class_weight=0.8)),
    ("drmm_tks", "mtple", dict(weight_decay=0.0, rerank_weight=0.5, class_weight=0.7)),
    ("drmm_tks", "bicut", dict(lr=1e-4, dropout=0.01)),
    ("bm25", "attncut", dict(batch_size=64, dropout=0.32503772565249145, weight_decay=0.0019306977288832496)),
    ("bm25", "mmoecut", dict(rerank_weight=0.2, class_weight=0.8)),
])
def test_conf_sections_override_the_command_line(dataset, model, expect):
    """Values as published with the reference's hyper_parameter_*.conf files."""
    run, args = _args(dataset_name=dataset, model_name=model)
    args = run.apply_conf(args)
    for k, v in expect.items():
        assert getattr(args, k) == pytest.approx(v), (k, getattr(args, k))


def test_missing_conf_section_raises_like_the_reference():
    run, args = _args(dataset_name="bm25", model_name="mtple")          # the reference's bm25 file has no [mtple_conf] either
    with pytest.raises(configparser.NoSectionError):
        run.apply_conf(args)
    run, args = _args(dataset_name="no_such_dataset", model_name="attncut")
    with pytest.raises(configparser.NoSectionError):
        run.apply_conf(args)


def test_scalar_log_uses_the_reference_tags(tmp_path):
    import run
    log = run.ScalarLog(str(tmp_path))
    log.add_scalar("train/loss_step", 0.5, 0)
    log.add_scalar("test/F1_epoch", 0.25, 3)
    log.close()
    rows = [json.loads(line) for line in open(os.path.join(tmp_path, "scalars.jsonl"))]
    assert rows == [{"tag": "train/loss_step", "value": 0.5, "step": 0}, {"tag": "test/F1_epoch", "value": 0.25, "step": 3}]
    run.ScalarLog(None).add_scalar("x", 1.0, 0)          # disabled: a no-op
