"""The table of golden cases: shared by tools/make_golden.py (which runs the REFERENCE on
them) and by the tests (which run the oracle and the HIP path on them).  Data only."""

SINGLE_CRITERIA = (
    [f"div_{d}_{m}_aug{a}" for d in ("js", "kl") for m in ("f1", "dcg") for a in (1, 0)]
    + [f"{c}_{m}" for m in ("f1", "dcg") for c in ("choopy", "attncutloss")]
)
FEW = ["div_js_f1_aug1", "div_kl_dcg_aug0", "choopy_f1"]
MT = ["mtcut_f1", "mtcut_dcg"]


def _case(tag, model, kwargs, batch, seq_len, n_feat, seed, criteria, grad_crit,
          gate_scale=None, w_r=0.5, w_c=0.5, pe_scale=None, x_noise=0.0):
    return dict(tag=tag, model=model, kwargs=kwargs, batch=batch, seq_len=seq_len, n_feat=n_feat,
                seed=seed, criteria=list(criteria), grad_crit=grad_crit, gate_scale=gate_scale,
                w_r=w_r, w_c=w_c, pe_scale=pe_scale, x_noise=x_noise)


MODEL_CASES = [
    _case("attncut_b5_s300", "AttnCut", {}, 5, 300, 3, 101, SINGLE_CRITERIA, "div_js_f1_aug1"),
    _case("attncut_b8_s100", "AttnCut", {}, 8, 100, 3, 102, FEW, "div_js_f1_aug1"),
    _case("attncut_b8_s200", "AttnCut", {}, 8, 200, 3, 103, FEW, "div_kl_dcg_aug0"),
    _case("attncut_b63_s300", "AttnCut", {}, 63, 300, 3, 104, ["div_js_f1_aug1"], "div_js_f1_aug1"),
    _case("choopy_b5_s300", "Choopy", {}, 5, 300, 1, 111, SINGLE_CRITERIA, "choopy_f1"),
    _case("choopy_b32_s300", "Choopy", {}, 32, 300, 1, 112, ["choopy_f1"], "choopy_f1"),
    _case("choopy_b6_s40", "Choopy", {"seq_len": 40}, 6, 40, 1, 113, FEW, "choopy_f1"),
    # many distinct cut positions per batch (small position encoding, unsorted list-specific scores): with the plain
    # recipe every list of a Choopy batch cuts at the same position, which makes "k identical" weak evidence
    _case("choopy_b24_s300_manyk", "Choopy", {}, 24, 300, 1, 118, ["choopy_f1", "div_js_f1_aug1"], "choopy_f1",
          pe_scale=0.05, x_noise=2.0),
    _case("choopy_b24_s300_closek", "Choopy", {}, 24, 300, 1, 120, ["choopy_f1"], "choopy_f1", pe_scale=0.05, x_noise=2.0),
    _case("mtattncut_t3_b5_s300", "MtAttnCut", {"num_tasks": 3}, 5, 300, 3, 122, MT, "mtcut_f1"),
    _case("mtattncut_t21_b5_s300", "MtAttnCut", {"num_tasks": 2.1}, 5, 300, 3, 120, MT, "mtcut_f1"),
    _case("mtattncut_t22_b5_s300", "MtAttnCut", {"num_tasks": 2.2}, 5, 300, 3, 121, MT, "mtcut_f1"),
    _case("mtattncut_t3_b8_s100", "MtAttnCut", {"num_tasks": 3}, 8, 100, 3, 127, MT, "mtcut_dcg"),
    _case("mtattncut_t3_b8_s200", "MtAttnCut", {"num_tasks": 3}, 8, 200, 3, 134, MT, "mtcut_f1"),
    _case("mtchoopy_t3_b5_s300", "MtChoopy", {"num_tasks": 3}, 5, 300, 1, 131, MT, "mtcut_f1"),
    _case("mtchoopy_t21_b4_s300", "MtChoopy", {"num_tasks": 2.1}, 4, 300, 1, 132, MT, "mtcut_f1"),
    _case("mtchoopy_t3_b12_s300_manyk", "MtChoopy", {"num_tasks": 3}, 12, 300, 1, 133, MT, "mtcut_f1", pe_scale=0.05, x_noise=2.0),
    _case("mmoecut_e3_t3_b5_s300", "MMOECut", {"num_experts": 3, "num_tasks": 3}, 5, 300, 3, 141, MT,
          "mtcut_f1", w_r=0.4, w_c=0.6),
    _case("mmoecut_e4_t21_b5_s300", "MMOECut", {"num_experts": 4, "num_tasks": 2.1}, 5, 300, 3, 142, MT,
          "mtcut_f1", w_r=0.4, w_c=0.6),
    _case("mmoecut_e4_t22_b5_s300", "MMOECut", {"num_experts": 4, "num_tasks": 2.2}, 5, 300, 3, 143, MT,
          "mtcut_f1", w_r=0.4, w_c=0.6),
    _case("mmoecut_e3_t3_b4_s300_refgate", "MMOECut", {"num_experts": 3, "num_tasks": 3}, 4, 300, 3, 144, MT,
          "mtcut_f1", gate_scale=1.0, w_r=0.4, w_c=0.6),
    _case("mmoecut_e3_t3_b6_s40", "MMOECut", {"num_experts": 3, "num_tasks": 3, "seq_len": 40}, 6, 40, 3, 145, MT,
          "mtcut_f1", w_r=0.4, w_c=0.6),
    # "next" row N4: the other two mixture-of-experts models of the reference
    _case("moecut_e3_t3_b5_s300", "MOECut", {"num_experts": 3, "num_tasks": 3}, 5, 300, 3, 151, MT,
          "mtcut_f1", w_r=0.4, w_c=0.6),
    _case("moecut_e4_t21_b6_s40", "MOECut", {"num_experts": 4, "num_tasks": 2.1, "seq_len": 40}, 6, 40, 3, 152, MT,
          "mtcut_dcg", w_r=0.4, w_c=0.6),
    _case("plecut_b5_s300", "PLECut", {}, 5, 300, 3, 161, MT, "mtcut_f1", w_r=0.4, w_c=0.6),
    _case("plecut_b6_s40", "PLECut", {"seq_len": 40}, 6, 40, 3, 162, MT, "mtcut_dcg", w_r=0.4, w_c=0.6),
]
CASE_BY_TAG = {c["tag"]: c for c in MODEL_CASES}


def make_criterion(losses_mod, name, case=None):
    """Build criterion `name` from a losses module exposing the reference's class names
    (the reference's utils.losses, oracle.losses, or the HIP package's utils.losses)."""
    parts = name.split("_")
    if parts[0] == "div":
        return losses_mod.DivLoss(metric=parts[2], div_type=parts[1], augmented=parts[3] == "aug1")
    if parts[0] == "choopy":
        return losses_mod.ChoopyLoss(metric=parts[1])
    if parts[0] == "attncutloss":
        return losses_mod.AttnCutLoss(metric=parts[1])
    if parts[0] == "mtcut":
        return losses_mod.MtCutLoss(metric=parts[1], rerank_weight=case["w_r"], classi_weight=case["w_c"],
                                    num_tasks=case["kwargs"].get("num_tasks", 3))
    raise KeyError(name)


# BiCut: outputs are (B,S,2); criteria are BiCutLoss metrics
BICUT_CASES = [
    dict(tag="bicut_b5_s300", kwargs={"input_size": 3}, batch=5, seq_len=300, n_feat=3, seed=171, criteria=["nci", "f1"], grad_crit="nci"),
    dict(tag="bicut_b8_s40_in5", kwargs={"input_size": 5}, batch=8, seq_len=40, n_feat=5, seed=172, criteria=["nci", "f1"], grad_crit="f1"),
]
