"""Training-batch side of the reference's datasets/phototourism_mask_grid_sample.py: the grid-sample strategy of
PhototourismDataset.__getitem__ (:241-275) with the flat ray / rgb buffers resident in HBM (a Brandenburg-Gate
training set at img_downscale 2 is a few GB -- it fits the 288 GB of one MI355X many times over) and the batch cut out
by one HIP gather kernel (crnerf_grid_sample_batch_f32) instead of host indexing + an H2D copy per step.

What is mirrored: the sampling arithmetic and its RNG stream (numpy seed per (epoch, idx), torch's global CPU generator
for scale / offsets) and the sample dict.  What is NOT here: COLMAP / image file reading (out of scope) --
the caller supplies all_rays[N,9], all_rgbs[N,3], all_imgs_wh[n_img,2] as the reference builds them (:139-200).
"""
from math import exp, sqrt

import numpy as np
import torch

from .. import ops


class GridSampleBatcher:
    def __init__(self, all_rays, all_rgbs, all_imgs_wh, batch_size=1024, scale_anneal=-1, min_scale=0.25, all_imgs=None):
        self.all_rays, self.all_rgbs = all_rays, all_rgbs              # device tensors
        self.all_imgs_wh = torch.as_tensor(all_imgs_wh).cpu().long()    # host: sizes drive host-side index arithmetic
        self.all_imgs = all_imgs                                        # optional list of whole images ('whole_img')
        self.batch_size, self.scale_anneal, self.min_scale = batch_size, scale_anneal, min_scale
        self.iterations = all_rays.shape[0] // batch_size               # :227
        self._offsets = torch.cat([torch.zeros(1, dtype=torch.long), (self.all_imgs_wh[:, 0] * self.all_imgs_wh[:, 1]).cumsum(0)])
        # every image's id (column 8 of its rays) on the HOST, read once: the training step indexes its appearance table with it
        # (train_mask_grid_sample.py:221 `ts[0]`) and must not wait for the device to learn it
        first = self._offsets[:-1].to(self.all_rays.device)
        self._image_ids = self.all_rays[first, 8].to(torch.int64).cpu().tolist() if self.all_rays.shape[1] > 8 and len(first) else []
        self._tables = {}

    def __len__(self):
        return self.iterations

    def _lin(self, size, side):
        key = (int(size), side)
        if key not in self._tables:   # torch.linspace on the HOST like the reference (:249-250), then one small upload.  `size` is
            # a 0-dim int64 tensor there, so the end point 1 - 1/size is rounded to fp32 before linspace sees it.
            self._tables[key] = torch.linspace(0, 1 - 1 / torch.as_tensor(int(size), dtype=torch.int64), side).to(self.all_rays.device)
        return self._tables[key]

    def draw(self, idx, current_epoch=0):
        """The host-side random choices of __getitem__ (:243-257), in the reference's order on the reference's generators
        (img_w / img_h stay 0-dim int64 tensors: the uniform bounds are fp32 tensor expressions there)."""
        np.random.seed(current_epoch * self.iterations + idx)
        sample_ts = int(np.random.randint(0, len(self.all_imgs_wh)))
        img_w, img_h = self.all_imgs_wh[sample_ts]
        if self.scale_anneal > 0:
            min_scale_cur = min(max(self.min_scale, 1. * exp(-(current_epoch * self.iterations + idx) * self.scale_anneal)), 0.9)
        else:
            min_scale_cur = self.min_scale
        scale = torch.Tensor(1).uniform_(min_scale_cur, 1.)
        h_offset = torch.Tensor(1).uniform_(0, (1 - scale.item()) * (1 - 1 / img_h))
        w_offset = torch.Tensor(1).uniform_(0, (1 - scale.item()) * (1 - 1 / img_w))
        return sample_ts, int(img_w), int(img_h), min_scale_cur, float(scale), float(h_offset), float(w_offset)

    def __getitem__(self, idx, current_epoch=0):
        sample_ts, img_w, img_h, min_scale_cur, scale, h_offset, w_offset = self.draw(idx, current_epoch)
        side = int(sqrt(self.batch_size))
        s = ops.grid_sample_batch(self.all_rays, self.all_rgbs, int(self._offsets[sample_ts]), img_w, img_h, side,
                                  self._lin(img_w, side), self._lin(img_h, side), scale, h_offset, w_offset)
        s.update({'whole_img': self.all_imgs[sample_ts] if self.all_imgs is not None else None, 'min_scale_cur': min_scale_cur,
                  'img_wh': self.all_imgs_wh[sample_ts]})
        if self._image_ids:
            s['image_id'] = self._image_ids[sample_ts]          # == int(s['ts'][0]), without the device round trip
        return s
