"""FIBERTransformerSS -- the drop-in boundary of the coarse-grained path, MI355X-native.

Mirrors coarse_grained/fiber/modules/fiber_module.py: constructor wiring (:27-179), infer() (:224-367, fused branch
:310-367, image-only :279-308, text-only :247-277), forward() (:431-471), training_step (:473-478),
configure_optimizers (:522).  Parameter names / shapes equal the reference's so a `fiber_pretrain.ckpt` state dict loads
(`load_path`, ITC queue keys dropped as at :141-146).  Captioning / ITC-queue / NLVR2 code is out of scope (SURVEY.md
section 2) and raises if requested.
"""
import os
import types

import torch
import torch.nn as nn

from .. import ops
from ..lightning import LightningModule
from . import fiber_utils, heads, objectives, roberta, swin_transformer
from .roberta import RobertaModel
from .swin_helpers import swin_adapt_position_encoding

_FORK_IMAGE = os.environ.get("FIBER_FORK_IMAGE", "1") != "0"      # A/B switch: 0 = autograd's own fan-in add for the image tokens of a fused step


@torch.no_grad()
def concat_all_gather(tensor):
    """fiber_module.py:12-24 (torch.distributed.all_gather, no gradient); the identity in a single-process run."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    out = [torch.empty_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(out, tensor.contiguous(), async_op=False)
    return torch.cat(out, dim=0)


class FIBERTransformerSS(LightningModule):
    def __init__(self, config):
        super().__init__()
        self.save_hyperparameters()
        self.config = config
        ln = config["loss_names"]
        for k in ("caption_mle", "caption_gold", "caption_cider", "nlvr2"):
            if ln.get(k, 0) > 0:
                raise NotImplementedError(f"loss '{k}' is outside the fused-backbone hot path built here")

        bert_config = types.SimpleNamespace(        # RobertaConfig defaults: layer_norm_eps = 1e-12
            vocab_size=config["vocab_size"], hidden_size=config["hidden_size"], layer_norm_eps=1e-12)

        # process-wide switch, like the reference's module globals below; ALWAYS set, so that a model built without the key does
        # not inherit the mode of the model built before it (default: env FIBER_RESIDUAL_DTYPE, else bf16)
        ops.set_residual_dtype(config.get("residual_dtype") or os.environ.get("FIBER_RESIDUAL_DTYPE", "bf16"))
        self.num_fuse_block = config["num_fuse_block"]
        self.num_text_layer = config["num_layers"]
        roberta.NUM_FUSE_BLOCK = swin_transformer.NUM_FUSE_BLOCK = self.num_fuse_block
        roberta.DIM_IMG = config["input_image_embed_size"]
        swin_transformer.DIM_TXT = config["input_text_embed_size"]      # reference typo kept: DIM_TEXT stays at its default
        swin_transformer.DIM_TEXT = config.get("swin_dim_text", 768)   # test-only knob for shrunken configs

        hs = config["hidden_size"]
        self.cross_modal_text_transform = nn.Linear(config["input_text_embed_size"], hs)
        self.cross_modal_image_transform = nn.Linear(config["input_image_embed_size"], hs)
        self.cross_modal_text_transform_itc = nn.Linear(config["input_text_embed_size"], hs)
        self.cross_modal_image_transform_itc = nn.Linear(config["input_image_embed_size"], hs)
        for m in (self.cross_modal_text_transform, self.cross_modal_image_transform,
                  self.cross_modal_text_transform_itc, self.cross_modal_image_transform_itc):
            m.apply(objectives.init_weights)

        if ln.get("itc", 0) > 0:                                         # ALBEF-style queues (fiber_module.py:56-67)
            self.temp = nn.Parameter(torch.ones([]) * 0.07)
            self.queue_size = qs = config.get("itc_queue_size", 4096)
            self.register_buffer("image_queue", torch.randn(hs, qs))
            self.register_buffer("text_queue", torch.randn(hs, qs))
            self.register_buffer("image_input_queue", torch.randn(qs, 3, config["image_size"], config["image_size"]))
            self.register_buffer("text_input_queue", torch.zeros(qs, config["max_text_len"], dtype=torch.long))
            self.register_buffer("text_input_mask_queue", torch.zeros(qs, config["max_text_len"], dtype=torch.long))
            self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
            self.register_buffer("queue_total", torch.zeros(1, dtype=torch.long))

        if "swin_arch" in config:                                        # test-only: explicit (embed_dim, depths, heads)
            dim, depths, nh = config["swin_arch"]
            self.vit_model = swin_transformer.SwinTransformer(img_size=config["image_size"], embed_dim=dim, depths=depths,
                                                              num_heads=nh, drop_path_rate=config.get("drop_path_rate", 0.1))
        else:
            kw = {"drop_path_rate": config["drop_path_rate"]} if "drop_path_rate" in config else {}
            self.vit_model = getattr(swin_transformer, config["vit"])(pretrained=config["pretrained_vit"], config=config, **kw)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        tover = {}
        if config["input_text_embed_size"] != 768 or config["num_layers"] != 12 or config["vocab_size"] != 50265:
            tover = dict(hidden_size=config["input_text_embed_size"], num_hidden_layers=config["num_layers"],
                         num_attention_heads=config["num_heads"], vocab_size=config["vocab_size"],
                         intermediate_size=config["input_text_embed_size"] * config["mlp_ratio"])
        if "max_position_embeddings" in config:
            tover["max_position_embeddings"] = config["max_position_embeddings"]
        if "text_dropout" in config:
            tover.update(hidden_dropout_prob=config["text_dropout"], attention_probs_dropout_prob=config["text_dropout"])
        self.text_transformer = RobertaModel.from_pretrained(config["tokenizer"], **tover)

        self.cross_modal_image_pooler = heads.Pooler(hs)
        self.cross_modal_text_pooler = heads.Pooler(hs)
        self.cross_modal_image_pooler.apply(objectives.init_weights)
        self.cross_modal_text_pooler.apply(objectives.init_weights)
        self.itc_pooler = config["itc_pooler"]
        if self.itc_pooler:
            self.cross_modal_image_pooler_itc = heads.Pooler(hs)
            self.cross_modal_text_pooler_itc = heads.Pooler(hs)
            self.cross_modal_image_pooler_itc.apply(objectives.init_weights)
            self.cross_modal_text_pooler_itc.apply(objectives.init_weights)

        if ln.get("mlm", 0) > 0:
            self.mlm_score = heads.MLMHead(bert_config)
            self.mlm_score.apply(objectives.init_weights)
        if ln.get("itm", 0) > 0:
            self.itm_score = heads.ITMHead(hs * 2)
            self.itm_score.apply(objectives.init_weights)
            self.rank_output = nn.Linear(hs, 1)
            self.rank_output.weight.data = self.itm_score.fc.weight.data[1:, :]
            self.rank_output.bias.data = self.itm_score.fc.bias.data[1:]

        if config["load_path"] != "" and not config.get("test_only", False):   # pre-trained -> downstream (:138-147)
            state_dict = self._read_checkpoint(config["load_path"])
            state_dict = swin_adapt_position_encoding(state_dict, before=config["resolution_before"],
                                                      after=config["image_size"])
            self.load_state_dict(state_dict, strict=False)

        if ln.get("vqa", 0) > 0:
            vs = config["vqav2_label_size"]
            self.vqa_classifier = heads.VQAClassifier(hs * 2, vs)
            self.vqa_classifier.apply(objectives.init_weights)

        fiber_utils.set_metrics(self)
        self.current_tasks = list()

        if config["load_path"] != "" and config.get("test_only", False):       # fine-tuned checkpoint, heads included (:173-180)
            self.load_state_dict(self._read_checkpoint(config["load_path"]), strict=False)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, image_feat, text_feat, image_input, text_input, text_input_mask):
        """fiber_module.py:181-222: the (all-gathered) batch overwrites the oldest queue slots, wrapping at queue_size;
        queue_total counts every sample ever enqueued (it bounds the hard-negative pool, objectives.py:143-155)."""
        image_feats, text_feats = concat_all_gather(image_feat), concat_all_gather(text_feat)
        image_input, text_input = concat_all_gather(image_input), concat_all_gather(text_input)
        text_input_mask = concat_all_gather(text_input_mask)
        n = image_feats.shape[0]
        slot = (int(self.queue_ptr) + torch.arange(n, device=image_feats.device)) % self.queue_size
        self.image_queue[:, slot] = image_feats.T.to(self.image_queue.dtype)
        self.text_queue[:, slot] = text_feats.T.to(self.text_queue.dtype)
        self.image_input_queue[slot] = image_input.to(self.image_input_queue.dtype)
        self.text_input_queue[slot] = text_input
        self.text_input_mask_queue[slot] = text_input_mask
        self.queue_ptr[0] = (int(self.queue_ptr) + n) % self.queue_size
        self.queue_total[0] = int(self.queue_total) + n

    @staticmethod
    def _read_checkpoint(path):
        state_dict = torch.load(path, map_location="cpu")["state_dict"]
        for key in ["image_queue", "text_queue", "queue_ptr", "queue_total", "image_input_queue", "text_input_queue",
                    "text_input_mask_queue"]:
            state_dict.pop(key, None)
        return state_dict

    # parameters that never receive a gradient for the configured losses (SURVEY.md section 7 "DDP unused parameters");
    # with ITC the image-only / text-only passes also exercise vit_model.norm, the *_itc transforms / poolers and layer 11's
    # output LayerNorm
    def unused_parameter_names(self):
        itc = self.config["loss_names"].get("itc", 0) > 0
        names = []
        for n, _ in self.named_parameters():
            if (n.startswith("text_transformer.pooler.") or n.startswith("rank_output.")
                    or ("crossattention_t2i.output.LayerNorm" in n)
                    or (not itc and (n.startswith("vit_model.norm.") or "_itc." in n))):
                names.append(n)
        last = self.num_text_layer - 1                  # layer 11 runs with last_norm=False on the fused path (:342)
        if not itc:
            names += [f"text_transformer.encoder.layer.{last}.output.LayerNorm.weight",
                      f"text_transformer.encoder.layer.{last}.output.LayerNorm.bias"]
        stage2 = self.vit_model.layers[2].blocks
        if len(stage2) < 8 + self.num_text_layer - self.num_fuse_block + 1:      # Swin-T quirk: text layers 6..9 never run
            for i in range(self.num_text_layer - self.num_fuse_block, 10):
                names += [n for n, _ in self.named_parameters() if n.startswith(f"text_transformer.encoder.layer.{i}.")]
            names += [n for n, _ in self.named_parameters()
                      if n.startswith("vit_model.layers.2.") and ("i2t" in n)]
        # alpha_t2i of layers without cross-attention is created but unused
        for i in range(self.num_text_layer):
            lyr = self.text_transformer.encoder.layer[i]
            if not hasattr(lyr, "crossattention_t2i") or f"text_transformer.encoder.layer.{i}.alpha_t2i" in names:
                names.append(f"text_transformer.encoder.layer.{i}.alpha_t2i")
        return sorted(set(names))

    def infer(self, batch, mask_text=False, mask_image=False, image_token_type_idx=1, img=None, text_only=False,
              image_only=False):
        if not text_only and img is None:
            imgkey = f"image_{image_token_type_idx - 1}" if f"image_{image_token_type_idx - 1}" in batch else "image"
            img = batch[imgkey][0]
        if not image_only:
            do_mlm = "_mlm" if mask_text else ""
            text_ids, text_labels, text_masks = batch[f"text_ids{do_mlm}"], batch[f"text_labels{do_mlm}"], batch["text_masks"]
        vit, txt = self.vit_model, self.text_transformer

        if text_only:
            text_embeds = txt.embeddings(input_ids=text_ids)
            ext = txt.get_extended_attention_mask(text_masks, text_masks.size(), text_embeds.device)
            for layer in txt.encoder.layer:
                text_embeds = layer(text_embeds, ext)[0]
            text_embeds = ops.linear(text_embeds, self.cross_modal_text_transform_itc.weight, self.cross_modal_text_transform_itc.bias)
            cls = (self.cross_modal_text_pooler_itc(text_embeds) if self.itc_pooler else text_embeds[:, 0]).float()
            cls = cls / cls.norm(dim=-1, keepdim=True)
            return {"text_feats": text_embeds, "image_feats": None, "cls_feats": cls, "text_labels": text_labels,
                    "text_ids": text_ids, "text_masks": text_masks, "image": None}

        if image_only:
            x = vit.patch_embed(img)
            for layer in vit.layers:
                x = layer(x)
            x = ops.layernorm(x, vit.norm.weight, vit.norm.bias, vit.norm.eps)
            x = ops.linear(x, self.cross_modal_image_transform_itc.weight, self.cross_modal_image_transform_itc.bias)
            avg = x.float().mean(1, keepdim=True)
            cls = (self.cross_modal_image_pooler_itc(avg) if self.itc_pooler else avg[:, 0]).float()
            cls = cls / cls.norm(dim=-1, keepdim=True)
            return {"text_feats": None, "image_feats": x, "cls_feats": cls, "text_labels": None, "text_ids": None,
                    "text_masks": None, "image": None}

        # ---- fused branch (fiber_module.py:310-367) ------------------------------------------------------------
        # The text stack below the first fusion block (embeddings + layers 0..5: 20k-row GEMMs, 80-240 tiles for 256 CUs)
        # does not depend on the image stack below it (stages 0-1 and stage-2 blocks 0..13, mostly HBM-bound kernels), so it
        # can be issued on a second HIP stream and joined where the reference first mixes the two (autograd replays every
        # node on its forward stream, so the backward halves overlap the same way).  ON by default since round 2
        # (config["overlap_text_stream"] = False or FIBER_NO_OVERLAP=1 turns it off): +2.3 % on the step (same-box A/B 338.3 ->
        # 330.6 ms at B=256).  In round 1 it was off because hipBLASLt's stream-K GEMMs (persistent grids whose workgroups wait
        # for each other) from the two streams could deadlock; every GEMM of the two stacks is a hand-written kernel now
        # (no inter-workgroup waits), the library is left with the heads, which run after the join.  Parameter gradients are
        # accumulated by autograd on the parameters' own stream behind event waits, so DDP's bucket hooks see finished
        # gradients.  Not used while a hipGraph is being captured.
        num_pre_text = self.num_text_layer - self.num_fuse_block

        def text_prefix():
            t = txt.embeddings(input_ids=text_ids)
            e = txt.get_extended_attention_mask(text_masks, text_masks.size(), t.device)
            for layer in txt.encoder.layer[:num_pre_text]:
                t = layer(t, e)[0]
            return t, e

        side = None
        if (self.config.get("overlap_text_stream", True) and not os.environ.get("FIBER_NO_OVERLAP")
                and text_ids.is_cuda and not torch.cuda.is_current_stream_capturing()):
            side = self._text_stream(text_ids)
        if side is not None:
            main = torch.cuda.current_stream(text_ids.device)
            object.__setattr__(self, "_main_stream", main)     # parallel.wrap_ddp's comm hook orders buckets behind both streams
            side.wait_stream(main)
            with torch.cuda.stream(side):
                text_embeds, ext = text_prefix()
        else:
            text_embeds, ext = text_prefix()

        image_embeds = vit.patch_embed(img)
        for layer in vit.layers[:2]:
            image_embeds = layer(image_embeds)

        # Fusion blocks: image block (reads the text tokens) and text layer (reads the image tokens) of one step are
        # independent of each other (fiber_module.py:327-346 evaluates both from the previous step's pair), so the text layer
        # runs on the second stream next to the much larger image block; two event waits per step keep the pair in lock step.
        prefix_only = self.config.get("overlap_text_stream", True) == "prefix" or bool(os.environ.get("FIBER_OVERLAP_PREFIX_ONLY"))

        def fused_step(blk, layer, image_embeds, text_embeds, **kw):
            nonlocal side
            if side is not None and prefix_only:             # join once, then single-stream
                main.wait_stream(side)
                text_embeds.record_stream(main)
                ext.record_stream(main)
                side = None
            # The image tokens feed the text layer's t2i keys / values AND the image block: the block reads an alias handed out by that projection,
            # whose dX GEMM then adds the block's input gradient in its epilogue (no autograd fan-in pass over [B*L, C]: ops._LinearPacked).
            sink = [] if (_FORK_IMAGE and self.training and torch.is_grad_enabled()) else None
            if side is None:                                 # same call order as the two-stream form below: the dropout /
                new_text = layer(text_embeds, ext, encoder_hidden_states=image_embeds, image_alias_sink=sink, **kw)[0]   # DropPath key counters
                return blk(sink[0] if sink else image_embeds, text_embeds, ext), new_text  # do not depend on the mode
            main.wait_stream(side)                           # text tokens of the previous step (produced on `side`)
            side.wait_stream(main)                           # image tokens of the previous step (produced on `main`)
            text_embeds.record_stream(main)
            image_embeds.record_stream(side)
            with torch.cuda.stream(side):
                new_text = layer(text_embeds, ext, encoder_hidden_states=image_embeds, image_alias_sink=sink, **kw)[0]
            return blk(sink[0] if sink else image_embeds, text_embeds, ext), new_text

        num_pre_block = 8 + num_pre_text
        for blk_cnt, blk in enumerate(vit.layers[2].blocks):
            if blk_cnt == num_pre_block and side is not None:
                ext.record_stream(main)
            if blk_cnt < num_pre_block:
                image_embeds = blk(image_embeds)
            else:
                image_embeds, text_embeds = fused_step(blk, txt.encoder.layer[blk_cnt - 8], image_embeds, text_embeds)
        if vit.layers[2].downsample is not None:
            image_embeds = vit.layers[2].downsample(image_embeds)

        for blk_cnt, blk in enumerate(vit.layers[3].blocks):
            image_embeds, text_embeds = fused_step(blk, txt.encoder.layer[blk_cnt + 10], image_embeds, text_embeds,
                                                   last_norm=(blk_cnt == 0))
        if vit.layers[3].downsample is not None:
            image_embeds = vit.layers[3].downsample(image_embeds)
        if side is not None:
            main.wait_stream(side)
            text_embeds.record_stream(main)
            ext.record_stream(main)

        text_embeds = ops.linear(text_embeds, self.cross_modal_text_transform.weight, self.cross_modal_text_transform.bias)
        image_embeds = ops.linear(image_embeds, self.cross_modal_image_transform.weight, self.cross_modal_image_transform.bias)
        cls_feats_text = self.cross_modal_text_pooler(text_embeds)
        avg_image_feats = image_embeds.float().mean(1, keepdim=True)
        cls_feats_image = self.cross_modal_image_pooler(avg_image_feats)
        cls_feats = torch.cat([cls_feats_text, cls_feats_image], dim=-1)
        return {"text_feats": text_embeds, "image_feats": image_embeds, "cls_feats": cls_feats, "text_labels": text_labels,
                "text_ids": text_ids, "text_masks": text_masks, "image": img}

    def _text_stream(self, ref):
        """Second HIP stream for the text prefix (one per module, created on first use)."""
        if not ref.is_cuda:
            return None
        st = getattr(self, "_side_stream", None)
        if st is None or st.device != ref.device:
            st = torch.cuda.Stream(device=ref.device)
            object.__setattr__(self, "_side_stream", st)
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:                            # parameters of the text stack accumulate on the second stream by design
                quiet(False)
        return st

    def forward(self, batch):
        ret = dict()
        if len(self.current_tasks) == 0:
            ret.update(self.infer(batch))
            return ret
        if "itc" in self.current_tasks:                       # task_pretrain_mlm_itm_itc (reference forward :437-451)
            fuse = ("mlm" in self.current_tasks and "itm" in self.current_tasks and self.config.get("fuse_mlm_itm", True))
            if "mlm" in self.current_tasks and not fuse:
                ret.update(objectives.compute_mlm(self, batch))
            ret_itc, image_neg, text_neg, text_mask_neg = objectives.compute_itc(self, batch, batch.get("itc_neg_override"))
            ret.update(ret_itc)
            if fuse:                                          # MLM + hard-negative ITM as one 4B-sample fused pass
                ret.update(objectives.compute_mlm_itm_hardneg_fused(self, batch, image_neg, text_neg, text_mask_neg))
            elif "itm" in self.current_tasks:
                ret.update(objectives.compute_itm_hardneg(self, batch, image_neg, text_neg, text_mask_neg))
        elif ("mlm" in self.current_tasks and "itm" in self.current_tasks and self.config.get("fuse_mlm_itm", True)):
            ret.update(objectives.compute_mlm_itm_fused(self, batch, batch.get("itm_labels_override")))
        else:
            if "mlm" in self.current_tasks:
                ret.update(objectives.compute_mlm(self, batch))
            if "itm" in self.current_tasks:
                ret.update(objectives.compute_itm(self, batch, batch.get("itm_labels_override")))
        if "vqa" in self.current_tasks:
            ret.update(objectives.compute_vqa(self, batch))
        return ret

    def training_step(self, batch, batch_idx):
        fiber_utils.set_task(self)
        ops.set_rng_step(int(self.global_step))           # dropout / DropPath keys advance with the optimizer step (resume-safe)
        output = self(batch)
        return sum([v for k, v in output.items() if "loss" in k])

    def training_epoch_end(self, outs):
        fiber_utils.epoch_wrapup(self)

    def validation_step(self, batch, batch_idx):
        fiber_utils.set_task(self)
        return self(batch)

    def validation_epoch_end(self, outs):
        fiber_utils.epoch_wrapup(self)

    def test_step(self, batch, batch_idx):
        """fiber_module.py:490-507: forward, then the per-task test record (VQA answers; captioning is out of scope)."""
        fiber_utils.set_task(self)
        output = self(batch)
        ret = dict()
        if self.hparams.config["loss_names"].get("vqa", 0) > 0:
            ret.update(objectives.vqa_test_step(self, batch, output))
        return ret

    def test_epoch_end(self, outs):
        """fiber_module.py:509-520: write the VQA submission file (rank shards merged by rank 0), then the epoch wrap-up."""
        model_name = self.hparams.config["load_path"].split("/")[-1][:-5]
        if self.hparams.config["loss_names"].get("vqa", 0) > 0:
            objectives.vqa_test_wrapup(outs, model_name)
        fiber_utils.epoch_wrapup(self)

    def configure_optimizers(self):
        return fiber_utils.set_schedule(self)
