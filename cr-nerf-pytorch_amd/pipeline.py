"""Orchestration around the hot path with the reference's call contracts, so that the
reference's drivers can be re-created without Lightning / imageio / torchvision:

    get_model(hparams)            train_mask_grid_sample.py:38-64   (module wiring, .cuda())
    load_ckpt / extract_model_state_dict   utils/__init__.py:67-88  (Lightning 'state_dict' + attribute-name prefixes)
    batched_inference(...)        eval.py:29-59 / appearance_modification_video.py:71-102 (ray-chunk loop, dict concat)
    decode_image(...)             eval.py:288-295 (feature -> [1,64,H,W] view -> decoder -> [H*W,3])
    render_frame(...)             one video frame: rays generated on the device, style from the appearance encoder
    TrainingSystem                NeRFSystem.decode / forward / training_step, train_mask_grid_sample.py:127-226, :268-290
                                  (command/train.sh's configuration: encode_a, encode_random, encode_c, use_mask)
"""
from collections import defaultdict

import torch

from .datasets.ray_utils import generate_rays
from .models.linearStyleTransfer import encoder_sameoutputsize, style_net
from .models.nerf import NeRF_sigma, PosEmbedding
from .models.rendering import render_rays_cross_ray


def get_model(hparams_, device="cuda"):
    """{'coarse', 'decoder'[, 'fine']} exactly as the reference builds them (encode_a=True configuration)."""
    xyz, dirs = 6 * hparams_.N_emb_xyz + 3, 6 * hparams_.N_emb_dir + 3
    models = {"coarse": NeRF_sigma(typ="coarse", args=hparams_, in_channels_xyz=xyz, in_channels_dir=dirs).to(device)}
    if not getattr(hparams_, "encode_a", True):
        raise NotImplementedError("crnerf_amd: only the encode_a=True decoder (style_net) is implemented")
    models["decoder"] = style_net(args=hparams_, residual_blocks=getattr(hparams_, "decoder_num_res_blocks", 2)).to(device)
    if hparams_.N_importance > 0:
        models["fine"] = NeRF_sigma("fine", args=hparams_, in_channels_xyz=xyz, in_channels_dir=dirs,
                                    encode_appearance=getattr(hparams_, "encode_a", True), in_channels_a=getattr(hparams_, "N_a", 48),
                                    encode_random=getattr(hparams_, "encode_random", True)).to(device)
    return models


def get_embeddings(hparams_):
    return {"xyz": PosEmbedding(hparams_.N_emb_xyz - 1, hparams_.N_emb_xyz), "dir": PosEmbedding(hparams_.N_emb_dir - 1, hparams_.N_emb_dir)}


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=(), trust_checkpoint=False):
    """utils/__init__.py:67-83.  The reference's checkpoints are Lightning files: callback / optimizer / scheduler state and an
    argparse.Namespace of hyper-parameters next to 'state_dict'.  They are read with torch's restricted unpickler
    (weights_only=True) with argparse.Namespace allow-listed -- enough for the files the reference's training loop writes
    (tests/golden/make_golden_trained.py) and for bare state_dict files, and it cannot execute code from the file.
    trust_checkpoint=True falls back to the unrestricted pickle loader the reference itself uses (torch 1.13's torch.load,
    utils/__init__.py:68) for files that hold other Python objects: only for files you wrote yourself."""
    import argparse
    import collections
    import pickle
    # what a Lightning checkpoint of the reference's trainer holds beside tensors: the hparams Namespace and ordinary containers
    allowed = [argparse.Namespace, collections.OrderedDict, collections.defaultdict]
    try:
        if hasattr(torch.serialization, "safe_globals"):                 # torch >= 2.5
            with torch.serialization.safe_globals(allowed):
                ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        else:                                                            # 2.4: the allow-list is process-wide
            if hasattr(torch.serialization, "add_safe_globals"):
                torch.serialization.add_safe_globals(allowed)
            ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not trust_checkpoint:
            raise RuntimeError("crnerf_amd: %s holds Python objects beyond tensors / containers / argparse.Namespace and the restricted "
                               "loader refused it (%s).  If the file is your own, pass trust_checkpoint=True to unpickle it without "
                               "restrictions (this can run code stored in the file)." % (ckpt_path, str(e).splitlines()[0])) from e
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    ckpt = ckpt.get("state_dict", ckpt)                      # Lightning checkpoint or a bare state_dict
    out = {}
    for key, value in ckpt.items():
        if not key.startswith(model_name):
            continue
        key = key[len(model_name) + 1:]
        if not any(key.startswith(p) for p in prefixes_to_ignore):
            out[key] = value
    return out


def load_ckpt(model, ckpt_path, model_name="model", prefixes_to_ignore=(), trust_checkpoint=False):
    state = model.state_dict()
    state.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore, trust_checkpoint))
    model.load_state_dict(state)


@torch.no_grad()
def batched_inference(models, embeddings, rays, ts, N_samples, N_importance, use_disp, chunk, white_back, **kwargs):
    """Ray-chunk loop; results concatenated per key (perturb = 0, noise_std = 0, test_time = True)."""
    results = defaultdict(list)
    for i in range(0, rays.shape[0], chunk):
        part = render_rays_cross_ray(models, embeddings, rays[i:i + chunk], ts[i:i + chunk] if ts is not None else None,
                                     N_samples, use_disp, 0, 0, N_importance, chunk, white_back, test_time=True, **kwargs)
        for k, v in part.items():
            results[k].append(v)
    return {k: torch.cat(v, 0) for k, v in results.items()}


@torch.no_grad()
def decode_image(models, results, H, W, a_embedded_from_img, key=None):
    if key is None:
        key = "feature_fine" if "feature_fine" in results else "feature_coarse"
    feature = results[key]                                           # [H*W, 64], already pixel-major
    grid = feature.t().reshape(1, feature.shape[1], int(H), int(W))  # the reference's two rearranges: a view
    rgb = models["decoder"](grid, a_embedded_from_img)               # [1,3,H,W]
    return rgb.reshape(3, int(H) * int(W)).t()                       # '1 n1 h w -> (h w) n1'


@torch.no_grad()
def render_frame(models, embeddings, enc_a, style_img, H, W, K, c2w, hparams_, near=0.0, far=5.0, chunk=32768, precision=None, a_emb=None):
    """One frame of the appearance-hallucination video path (appearance_modification_video.py:224-262):
    style image -> appearance embedding, camera -> rays on the device, render, decode.  Returns [H,W,3] in [0,1].
    a_emb: a precomputed enc_a(style_img) -- the reference encodes the style image ONCE per video (:216-222), so frame loops
    pass it in instead of re-running the encoder per frame."""
    if a_emb is None:
        a_emb = enc_a(style_img)
    rays = generate_rays(H, W, K, c2w, near, far, device=style_img.device)
    res = batched_inference(models, embeddings, rays, None, hparams_.N_samples, hparams_.N_importance, hparams_.use_disp,
                            chunk, False, args=hparams_, a_embedded_from_img=a_emb, precision=precision)
    return decode_image(models, res, H, W, a_emb).reshape(int(H), int(W), 3).clamp(0, 1)


class _Branches:
    """Independent chains of a step on their own streams: run(k, fn) enqueues fn on stream k (which first waits for everything the calling stream
    has enqueued so far -- the first time it is used in this step; later calls on the same stream just follow), join() makes the calling stream
    wait for all of them.  streams = None: everything runs inline on the calling stream."""

    def __init__(self, streams):
        self.streams, self.used = streams, set()
        self.main = torch.cuda.current_stream() if streams is not None else None

    def run(self, k, fn):
        if self.streams is None:
            return fn()
        st = self.streams[k]
        if k not in self.used:
            st.wait_stream(self.main)
            self.used.add(k)
        torch.cuda.set_stream(st)             # (not `with torch.cuda.stream(st)`: the context manager costs ~20 us of host time per use, and this
        try:                                  # step is host-bound)
            return fn()
        finally:
            torch.cuda.set_stream(self.main)

    def join(self):
        for k in self.used:
            self.main.wait_stream(self.streams[k])


class TrainingSystem:
    """The training-side orchestration of the reference's NeRFSystem without Lightning: same dict keys, same order of
    operations, autograd through the HIP twins (models/rendering.py grad path, autograd.py) and the fused loss."""

    def __init__(self, hparams_, models=None, embeddings=None, enc_a=None, device="cuda", ray_parallel_group=False):
        """ray_parallel_group: False = every process trains on its own batches (the reference's DDP; all-reduce the
        gradients with parallel.allreduce_gradients); a torch.distributed group (None = the default group) = the ranks of
        the group split EVERY batch's rays between them (parallel.GatherRays) and sync_gradients() must follow backward."""
        from .losses import loss_dict
        self.ray_group = ray_parallel_group
        self.hparams_ = hparams_
        self.loss = loss_dict['crnerf'](hparams_, coef=1)                                   # :74
        self.models = models or get_model(hparams_, device)
        self.embeddings = embeddings or get_embeddings(hparams_)
        self.enc_a = enc_a if enc_a is not None else encoder_sameoutputsize(out_channel=hparams_.nerf_out_dim).to(device)   # :95
        self.enc_cont = encoder_sameoutputsize(out_channel=hparams_.nerf_out_dim).to(device) if getattr(hparams_, "encode_c", False) else None   # :84
        self.embedding_a_list = [None] * getattr(hparams_, "N_vocab", 1500)                 # :98
        self.models_to_train = list(self.models.values()) + [self.enc_a] + ([self.enc_cont] if self.enc_cont is not None else [])   # :84-97
        self.implicit_mask = None
        if getattr(hparams_, "use_mask", False):                                            # :113-115
            from .models.lightweight_seg import Context_Guided_Network
            self.implicit_mask = Context_Guided_Network(classes=1, M=2, N=2, input_channel=3).to(device)
            self.models_to_train += [self.implicit_mask]
        self.global_step = 0
        self.training = True
        # the modules a step calls several times (enc_a x3, decoder x3, enc_cont x2) sum their parameter gradients in ONE multi-tensor add at the
        # end of backward instead of one AccumulateGrad add per tensor and use (autograd.deferred_param_grads).  Off where torch DDP may be
        # listening on AccumulateGrad (a process group exists and this system does not shard rays itself); set the attribute to override.
        # Decided at every training_step (a process group created, or the modules wrapped in DDP, AFTER this constructor must still
        # switch it off); assign True / False to `fused_grad_accumulation` to override.
        self._ray_parallel_requested = ray_parallel_group is not False
        self._fused_grad_override = None
        # ray-parallel mode: the encoder passes over the re-rendered images run as row bands, one per rank (round 6); False = replicated, as up to round 5
        self.shard_encoders = True
        self.shard_decodes = True               # ... and the four decodes over each rank's own pixels (DecodeShardedFn); False: features gathered, decodes replicated
        self.after_render = None                # callable(results) between the render (+ feature gather) and the decodes, or None
        import os
        # the decode / encoder chains of a step side by side on four streams (forward's comment); False / CRNERF_BRANCH_STREAMS=0: one stream
        self.branch_streams = os.environ.get("CRNERF_BRANCH_STREAMS", "1") != "0"
        self._streams = {}

    @property
    def fused_grad_accumulation(self):
        if self._fused_grad_override is not None:
            return self._fused_grad_override
        return self._ray_parallel_requested or not (torch.distributed.is_available() and torch.distributed.is_initialized())

    @fused_grad_accumulation.setter
    def fused_grad_accumulation(self, on):
        self._fused_grad_override = None if on is None else bool(on)

    def parameters(self):
        return [p for m in self.models_to_train for p in m.parameters()]

    def train(self, mode=True):
        """LightningModule.train()/eval() over every trained module (BatchNorm of the mask network, packed-weight caches)."""
        for m in self.models_to_train:
            m.train(mode)
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _draw(self, seen):
        """random.choice(seen) of train_mask_grid_sample.py:165-167.  In ray-parallel mode every rank must decode 'fine_random'
        with the SAME style (all ranks evaluate one full-batch loss, parallel.GatherRays), so the draw comes from a generator
        seeded by the step counter instead of the process-local `random` state."""
        import random
        if self.ray_group is False:
            return random.choice(seen)
        return random.Random(0x5EED + self.global_step).choice(seen)

    def sync_gradients(self):
        """Ray-parallel mode, after loss.backward(): MLP gradients summed over the ranks, replicated modules averaged."""
        from .parallel import sync_ray_parallel_gradients
        sharded = [m for k, m in self.models.items() if k in ("coarse", "fine")]
        sync_ray_parallel_gradients(sharded, [m for m in self.models_to_train if m not in sharded], self.ray_group)

    def _branch_streams(self, device):
        key = torch.device(device).index
        if key not in self._streams:
            self._streams[key] = [torch.cuda.Stream(device=device) for _ in range(6)]
        return self._streams[key]

    def _encode(self, enc, image):
        """An encoder pass over a re-rendered image (:219, :223-224).  Ray-parallel mode: the ranks split the image into row bands
        (parallel.encode_banded -- every rank holds the whole image, the decoder being replicated); anywhere else, or when the image does not
        split evenly over the ranks, the whole pass."""
        if self.ray_group is not False and self.shard_encoders and self.training:
            from .parallel import encode_banded
            out = encode_banded(enc, image, self.ray_group)
            if out is not None:
                return out
        return enc(image)

    def decode(self, results, type, **kwargs):                                              # :127-149
        feature = results['feature_' + type] if type != 'content' else results['feature_fine']
        H, W = int(kwargs['H']), int(kwargs['W'])
        grid = None if getattr(self, "_sharded_decode", None) else feature.t().reshape(1, feature.shape[-1], H, W)   # 'n1 n3 -> n3 n1' then ' n3 (h w) -> 1 n3 h w'
        sharded = getattr(self, "_sharded_decode", None)      # ray-parallel training: (group, rays of the whole batch) -- `feature` is this rank's block
        if type == "content":                                                               # decoder only, no appearance transfer (:145-148)
            if sharded:
                from .parallel import decode_sharded_train
                results['rgb_content_img'] = decode_sharded_train(self.models['decoder'], feature, None, sharded[1], sharded[0], content_only=True).view(1, 3, H, W)
            else:
                results['rgb_content_img'] = self.models['decoder'](grid, None, type="content")
            results['rgb_content'] = None
            return results
        style = kwargs['a_embedded_random'] if type == "fine_random" else kwargs['a_embedded_from_img']
        if sharded:
            from .parallel import decode_sharded_train
            rgbs_pred = decode_sharded_train(self.models['decoder'], feature, style, sharded[1], sharded[0]).view(1, 3, H, W)
        else:
            rgbs_pred = self.models['decoder'](grid, style)
        if type == "fine":
            results['rgb_fine_img'] = rgbs_pred
        if type != "fine_random":                                                           # fine_random is rearranged later (:221)
            rgbs_pred = rgbs_pred.reshape(3, H * W).t()                                     # ' 1 n1 h w -> (h w) n1' (a strided view)
        results['rgb_' + type] = rgbs_pred
        return results

    def forward(self, rays, ts, whole_img, W, H, rgb_idx=None, hw_whole=None, val_mode=False, image_id=None):   # :151-226
        """val_mode=True (validation_step, :362): the transient mask is the whole interpolated image (no rgb_idx gather, :174-175)
        and the reference switches to 2,048-ray chunks (:181-182) -- a memory bound only: rays are independent, so the chunk
        size never changes the result and the renderer's own chunking is kept."""
        hp = self.hparams_
        results = defaultdict(list)
        kwargs = {'args': hp}
        whole_img = (whole_img + 1) / 2                                                     # [-1,1] -> [0,1]  :156
        # The photo's two consumers -- the appearance encoder and the mask network -- need nothing from the renderer and the renderer nothing
        # from them: with branch streams they run beside it, forward and (a node's backward runs on its forward's stream) backward, where the
        # mask network's ~90 small launches slot in between the workgroups of the MLP's big kernels.  Joined before the decodes / the loss.
        use_br = self.branch_streams and self.training and torch.is_grad_enabled() and self.ray_group is False
        pre = _Branches(self._branch_streams(rays.device)[4:6] if use_br else None)
        kwargs['a_embedded_from_img'] = pre.run(0, lambda: self.enc_a(whole_img))
        if hp.encode_random:
            seen = [k for k, v in enumerate(self.embedding_a_list) if v is not None]
            kwargs['a_embedded_random'] = kwargs['a_embedded_from_img'] if len(seen) == 0 else self.embedding_a_list[self._draw(seen)]
        if self.implicit_mask is not None:                                                  # :170-176
            from .models.lightweight_seg import mask_at_pixels
            if hw_whole is None or (rgb_idx is None and not val_mode):
                raise ValueError("crnerf_amd: use_mask needs the batch's rgb_idx and the full-resolution image size (hw_whole)")

            def mask_chain():
                pred_mask = self.implicit_mask(whole_img)
                # interpolate(pred_mask, hw_whole) -> '(h w) n' -> [rgb_idx], evaluated only at the batch's pixels (all of them in val_mode)
                return mask_at_pixels(pred_mask, hw_whole, None if val_mode else rgb_idx.reshape(-1))
            kwargs['mask_embedded_from_img'] = pre.run(1, mask_chain)
        kwargs["H"], kwargs["W"] = H, W
        B = rays.shape[0]
        image_id = int(ts[0]) if image_id is None else int(image_id)   # (int(ts[0]) waits for the device; the batcher knows it on the host)
        lo = 0
        if self.ray_group is not False:                                                     # this rank's block of the batch
            from .parallel import gather_rays, shard_rays
            rays, (lo, hi) = shard_rays(rays, self.ray_group)
            ts = ts[lo:hi]
        ray_chunk = max(int(hp.chunk), 1 << 16)   # the reference's 8,192-ray chunks (:185-197) only bound its memory; rays are independent
        for i in range(0, rays.shape[0], ray_chunk):
            # rng_ray_offset: ray i of this rank's shard is ray lo + i of the batch -- every rank seeds alike (one seed per step), so without it ray j of
            # EVERY shard would get the same jitter / noise; with it an N-rank step draws exactly what the 1-rank step draws
            part = render_rays_cross_ray(self.models, self.embeddings, rays[i:i + ray_chunk], ts[i:i + ray_chunk], hp.N_samples, hp.use_disp,
                                         hp.perturb, hp.noise_std, hp.N_importance, hp.chunk, False, rng_ray_offset=lo + i, **kwargs)
            for k, v in part.items():
                results[k] += [v]
        for k, v in results.items():
            results[k] = v[0] if len(v) == 1 else torch.cat(v, 0)    # (one chunk is the rule here: cat of a single tensor would copy it)
        self._sharded_decode = None
        if self.ray_group is not False:
            import torch.distributed as dist
            ws = dist.get_world_size(self.ray_group)
            if self.shard_decodes and self.training and torch.is_grad_enabled() and B % ws == 0 and B // ws > 0:
                # round 6: the decodes stay sharded -- every rank decodes its own pixels around the two all-reduces of the decoder's statistics and
                # RGB is all-gathered (parallel.decode_sharded_train); the feature rows are NOT gathered, their gradient never leaves the rank
                self._sharded_decode = (self.ray_group, B)
            else:      # the decoder is cross-ray: every rank gets the whole feature grid (weights_* / depth_* stay local)
                for k in ("feature_coarse", "feature_fine"):
                    if k in results:
                        results[k] = gather_rays(results[k], B, self.ray_group)
                if "feature_fine_random" in results:
                    results["feature_fine_random"] = results["feature_fine"]
        pre.join()                             # the appearance embedding (the decodes read it) and the mask (the loss does)
        if self.after_render is not None:      # measurement hook (bench.py --workload configs3): the boundary between the ray-sharded part and the rest
            self.after_render(results)
        # The four decodes (:205-218) and the three encoder passes over the re-rendered images (:219, :223-224) are four independent chains of small
        # kernels -- coarse | fine -> enc_cont | content -> enc_cont | fine_random -> enc_a -- each a few dozen launches that use a few CUs for a
        # few microseconds.  Round 6: one HIP stream per chain (self.branch_streams), so the chains run side by side instead of one after the
        # other, forward and -- a node's backward runs on its forward's stream -- backward.  Same kernels, same arithmetic, same results.
        br = _Branches(self._branch_streams(rays.device)[:4] if use_br else None)
        br.run(0, lambda: self.decode(results, "coarse", **kwargs))
        if hp.N_importance > 0:
            br.run(1, lambda: self.decode(results, "fine", **kwargs))
        if getattr(hp, "encode_c", False):
            br.run(2, lambda: self.decode(results, "content", **kwargs))                     # :207-208
        if self.implicit_mask is not None:
            results['out_mask'] = kwargs['mask_embedded_from_img']                          # :210-211
        results['a_embedded'] = kwargs['a_embedded_from_img']
        results['whole_img'] = whole_img
        if hp.encode_random:
            results['a_embedded_random'] = kwargs['a_embedded_random']

            def random_chain():
                self.decode(results, "fine_random", **kwargs)
                results['a_embedded_random_rec'] = self._encode(self.enc_a, results['rgb_fine_random'])       # :219
                results['rgb_fine_random'] = results['rgb_fine_random'].reshape(3, int(H) * int(W)).t()
            br.run(3, random_chain)
            self.embedding_a_list[image_id] = kwargs['a_embedded_from_img'].clone().detach()
        if getattr(hp, "encode_c", False):                                                  # :222-224
            br.run(1, lambda: results.__setitem__('content_with_a_embed', self._encode(self.enc_cont, results['rgb_fine_img'])))
            br.run(2, lambda: results.__setitem__('content_wo_a_embed', self._encode(self.enc_cont, results['rgb_content_img'])))
        br.join()
        return results

    def training_step(self, batch):                                                         # :268-290
        rays, ts, rgbs = batch['rays'], batch['ts'], batch['rgbs']
        side = int(round(rays.shape[0] ** 0.5))
        hw_whole = None
        if batch.get('img_wh') is not None:
            w_whole, h_whole = (int(v) for v in batch['img_wh'])                            # :272
            hw_whole = (h_whole, w_whole)
        from .autograd import deferred_param_grads
        with deferred_param_grads(self.fused_grad_accumulation):
            results = self.forward(rays, ts, batch['whole_img'], side, side, batch.get('rgb_idx'), hw_whole, image_id=batch.get('image_id'))
        loss_d, annealing = self.loss(results, rgbs, self.hparams_, self.global_step)
        loss = loss_d.total() if hasattr(loss_d, "total") else sum(l for l in loss_d.values())   # (:286; one reduction of the kernel's 7-vector)
        self.global_step += 1
        return loss, loss_d, results

    @torch.no_grad()
    def validation_step(self, batch, batch_nb=0):                                           # :339-402
        """One whole validation image (PhototourismDataset split='val' hands over every ray of the image, `img_wh` = its size):
        forward in val_mode, the training loss terms, PSNR of the finest decode (metrics.py:12-13).  Returns the reference's log
        dict -- 'val_loss', the loss terms, 'val_psnr' -- plus 'results' for callers that want the images (the reference sends
        img_gt / prediction / random-appearance prediction / mask to wandb here, :371-391; its 'val_ssim' is kornia's, out of scope)."""
        rays, ts, rgbs = batch['rays'].squeeze(), batch['ts'].squeeze(), batch['rgbs'].squeeze()
        W, H = (int(v) for v in torch.as_tensor(batch['img_wh']).reshape(-1)[:2])          # :346-347
        was_training = self.training
        self.eval()                                                                         # Lightning runs validation in eval mode
        try:
            results = self.forward(rays, ts, batch['whole_img'], W, H, batch.get('rgb_idx'), hw_whole=(H, W), val_mode=True)
        finally:
            self.train(was_training)
        loss_d, _ = self.loss(results, rgbs, self.hparams_, self.global_step)
        log = {'val_loss': sum(l for l in loss_d.values())}
        log.update(loss_d)
        typ = 'fine' if 'rgb_fine' in results else 'coarse'
        log['val_psnr'] = -10.0 * torch.log10(((results['rgb_%s' % typ] - rgbs) ** 2).mean())
        log['results'] = results
        return log


__all__ = ["get_model", "get_embeddings", "load_ckpt", "extract_model_state_dict", "batched_inference", "decode_image", "render_frame",
           "encoder_sameoutputsize", "TrainingSystem"]
