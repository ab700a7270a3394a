"""Default configuration dict of the coarse-grained path (values from the reference's sacred config,
coarse_grained/fiber/config.py:21-92; sacred itself is not needed to build the dict)."""
import copy


def _loss_names(d):
    ret = {"itm": 0, "itc": 0, "mlm": 0, "vqa": 0, "nlvr2": 0, "caption_mle": 0, "caption_gold": 0, "caption_cider": 0}
    ret.update(d)
    return ret


DEFAULTS = dict(
    exp_name="fiber", seed=0, loss_names=_loss_names({"itm": 1, "mlm": 1}), batch_size=4096,
    image_size=384, vit="swin_base_patch4_window12_384_in22k", image_only=False, draw_false_image=1,
    input_image_embed_size=1024, resolution_before=384, pretrained_vit=False,
    vqav2_label_size=3129, max_text_len=40, tokenizer="roberta-base", vocab_size=50265, whole_word_masking=False,
    mlm_prob=0.15, draw_false_text=0, input_text_embed_size=768,
    hidden_size=768, num_heads=12, num_layers=12, mlp_ratio=4, drop_rate=0.1, num_fuse_block=6, itc_pooler=True,
    optim_type="adamw", learning_rate=1e-5, weight_decay=0.01, decay_power=1, max_epoch=100, max_steps=100000,
    warmup_steps=10000, end_lr=0, lr_mult_head=5, lr_mult_cross_modal=5,
    get_recall_metric=False, get_recall_metric_itc=True, cider_path=None,
    resume_from=None, fast_dev_run=False, val_check_interval=1.0, test_only=False,
    data_root="", log_dir="result", per_gpu_batchsize=0, num_gpus=8, num_nodes=1, load_path="", num_workers=8, precision=32,
    # not a reference key: storage type of the residual streams of both backbones -- "bf16" (rounded at every block) or "fp32"
    # (fiber_amd/ops.py "fp32 residual stream"; the reference's fp32 run keeps them in fp32).  None = FIBER_RESIDUAL_DTYPE or bf16.
    residual_dtype=None,
    # not a reference key: True = MLM head only on the rows that carry a label (objectives._mlm_head): same loss, gradients and accuracy,
    # but ret["mlm_logits"] is then [n_labelled, V]; the default keeps the reference's full [B, S, V] head
    mlm_compact_rows=False,
)


def make_config(**over):
    c = copy.deepcopy(DEFAULTS)
    for k, v in over.items():
        if k == "loss_names":
            c[k] = _loss_names(v)
        else:
            c[k] = v
    return c


# Named task configs of the path built here (reference config.py:95-110 pre-training with / without ITC, :134-150 VQAv2).
NAMED = {
    "task_pretrain_mlm_itm": dict(
        exp_name="mlm_itm", loss_names={"itm": 1, "mlm": 1}, draw_false_image=1, batch_size=4096, max_steps=100000,
        warmup_steps=0.1, whole_word_masking=False, learning_rate=1e-5, lr_mult_cross_modal=5, lr_mult_head=5),
    "task_pretrain_mlm_itm_itc": dict(
        exp_name="mlm_itm_itc", loss_names={"itm": 1, "mlm": 1, "itc": 1}, draw_false_image=0, batch_size=4096,
        max_steps=100000, warmup_steps=0.1, whole_word_masking=False, learning_rate=1e-5, lr_mult_cross_modal=5,
        lr_mult_head=5),
    "task_finetune_vqa": dict(
        exp_name="finetune_vqa", loss_names={"vqa": 1}, batch_size=512, max_epoch=10, max_steps=None, warmup_steps=0.1,
        learning_rate=2e-5, lr_mult_cross_modal=5, lr_mult_head=50, max_text_len=50, image_size=576,
        pretrained_vit=False, draw_false_image=0),
}


def named_config(name, **over):
    """make_config(**NAMED[name], **over): the sacred `with <name> key=value` command line without sacred."""
    kw = dict(NAMED[name])
    kw.update(over)
    return make_config(**kw)
