"""HIPT_4K host mirror (reference: ``HIPT_4K/hipt_4k.py:31-118, 308-330``).

``HIPT_4K.forward`` keeps the reference contract — ``x [1,3,W',H'] -> [1,192]`` on ``device4k`` —
but the whole region is processed by ONE library call when both ViTs share a device: centre crop,
patchify (fused into the patch-embedding addressing), ViT-256 over all 256x256 patches, the
[CLS] grid handed to ViT-4K on device (the reference's GPU->CPU->GPU hop of cls256,
hipt_4k.py:70,74, is gone), ViT-4K.
"""
from __future__ import annotations

import torch

from . import _native as N
from . import functional as Fn
from .hipt_model_utils import get_vit256, get_vit4k


def center_crop_box(size: int, crop: int) -> int:
    """torchvision.transforms.CenterCrop offset: int(round((size - crop) / 2.0))."""
    return int(round((size - crop) / 2.0))


class HIPT_4K(torch.nn.Module):
    """Hierarchical ViT feature extractor for [256x256]-patch regions (hipt_4k.py:31-46)."""

    def __init__(self, model256_path: str = '../Checkpoints/vit256_small_dino.pth',
                 model4k_path: str = '../Checkpoints/vit4k_xs_dino.pth',
                 device256=torch.device('cuda:0'), device4k=torch.device('cuda:1'), compute_dtype=None):
        super().__init__()
        self.model256 = get_vit256(pretrained_weights=model256_path).to(device256)
        self.model4k = get_vit4k(pretrained_weights=model4k_path).to(device4k)
        self.device256 = torch.device(device256)
        self.device4k = torch.device(device4k)
        self.chunk = 0  # patches per ViT-256 pass (0 = library default)
        # > 1: a batch of regions is cut into that many parts, each run on its own HIP stream with its own workspace.
        # Every kernel of the path occupies whole CUs (one persistent workgroup each), so the parts do not share CUs:
        # the second stream's kernels fill the CUs the first one's kernel frees in its last, partial round of tiles.
        # `streams` is the MOST a call may use; how many it does use depends on its size (`_parts`, round 6): splitting pays only once every part
        # still fills the chip for many rounds of tiles.  Default 2.
        self.streams = 2
        # ... and a batch of ONE region (the reference's call pattern) as patch ranges over this many streams (_run_patch_split).  Default 2
        # (round 6): one region is 514 row tiles of the fused MLP = two rounds on 256 CUs + 2 tiles, a third pass over the weights for 0.4 % of the
        # rows; two half-size streams (257 tiles each) run their leftover tile beside the other stream's full round.  tools/batch1_bench.py on four
        # boxes of the pool in round 6: 1 stream 224-233 regions/s, 2 streams 236-249 (+6-7 % on every one), 3 / 4 streams 186-203.  (Round 5 measured
        # 2 streams behind 1 on some boxes, before ViT-4K's launches were cut; the same bits either way: tests.)
        self.patch_streams = 2
        self._side_streams = {}
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, name: str):
        self.model256.set_compute_dtype(name)
        self.model4k.set_compute_dtype(name)
        return self

    # ---- hipt_4k.py:308-330 -------------------------------------------------------------
    def prepare_img_tensor(self, img: torch.Tensor, patch_size=256):
        if self._interleaved(img):  # uint8 [R, W, H, 3] (decoded RGB tiles): same crop on the two middle axes
            b, w, h, c = img.shape
        else:
            b, c, w, h = img.shape
        W, H = w - w % patch_size, h - h % patch_size
        if (W, H) != (w, h):
            t, l = center_crop_box(w, W), center_crop_box(h, H)
            img = img[:, t:t + W, l:l + H, :] if self._interleaved(img) else img[:, :, t:t + W, l:l + H]
        return img, w // patch_size, h // patch_size

    @staticmethod
    def _interleaved(img: torch.Tensor) -> bool:
        return img.dtype == torch.uint8 and img.dim() == 4 and img.shape[-1] == 3 and img.shape[1] != 3

    def _same_device(self) -> bool:
        return self.model256.weight_device == self.model4k.weight_device

    def _run(self, x: torch.Tensor, want_cls256: bool):
        batch, w_256, h_256 = self.prepare_img_tensor(x)
        if w_256 == 0 or h_256 == 0:
            raise ValueError(f"region {tuple(x.shape)} is smaller than one 256x256 patch")
        nreg = batch.shape[0]  # the reference takes 1 (hipt_4k.py:73); R > 1 regions are independent -> stacked
        # (weight_device, not next(parameters()): a nn.DataParallel replica has no parameters -- the reference wraps the
        #  model whenever it sees more than one GPU, extract_features_fp.py:217-218 -- and .to() may have moved the ViTs
        #  since the constructor recorded device256 / device4k)
        d256, d4k = self.model256.weight_device, self.model4k.weight_device
        u8 = batch.dtype == torch.uint8  # raw RGB bytes: ToTensor + Normalize(0.5, 0.5) happen on the device
        hwc = self._interleaved(batch)
        region = batch.to(d256, non_blocking=True).detach()
        region = region.contiguous() if u8 else region.float().contiguous()
        N.require_cuda(region, "HIPT_4K")
        if region.data_ptr() % 16:  # a view that starts off the 16-byte grid (the kernels load 16 bytes at a time): one aligned copy
            region = region.clone()
        per = w_256 * h_256
        nseq = nreg * per
        W, H = (region.shape[1], region.shape[2]) if hwc else (region.shape[2], region.shape[3])
        if u8 and d256 != d4k:  # two-device placement: normalise here, then the float path below
            m256 = self.model256
            buf = torch.empty((nreg, 3, W, H), dtype=torch.float32, device=d256)
            N.call("hipt_u8_normalize", N.ptr(region), int(hwc), nreg, W * H, N.ptr(buf), N.HIPT_F32, N.stream_ptr(d256))
            region, u8 = buf, False
        if d256 == d4k:
            m256, m4k = self.model256, self.model4k
            pk256 = m256._packed_for(m256._pos_for(256, 256, 256))
            # fewer regions than streams: the patches are spread over the streams (decided before anything is allocated or packed for
            # the whole-region path: that path's `out` and ViT-4K image are not used there)
            split = int(self.patch_streams) if nreg == 1 else 1  # (only a call of ONE region is cut by patches; groups of regions: _parts)
            while split > 1 and nseq < 32 * split:
                split -= 1  # (small regions: fewer ranges)
            if split > nreg and nseq >= 32 * split:
                return self._run_patch_split(region, u8, hwc, nreg, w_256, h_256, W, H, pk256, d256, want_cls256, split)
            pk4k = m4k._packed_for(m4k._pos_for(per, w_256, h_256))
            out = torch.empty((nreg, pk4k.w.dim), dtype=torch.float32, device=d4k)
            cls256 = torch.empty((nseq, pk256.w.dim), dtype=torch.float32, device=d256) if want_cls256 else None

            def launch(lo, hi, slot):
                n = hi - lo
                sub_cls = cls256[lo * per:hi * per] if cls256 is not None else None
                if u8:
                    need = N.lib().hipt_hipt4k_u8_workspace_bytes(pk256.ref, pk4k.ref, n, w_256, h_256, self.chunk)
                    ws = Fn.workspace(d256, need, slot)
                    N.call("hipt_hipt4k_forward_u8", pk256.ref, pk4k.ref, N.ptr(region[lo:hi]), int(hwc), n, W, H, self.chunk,
                           N.ptr(sub_cls), N.ptr(out[lo:hi]), N.ptr(ws), ws.numel(), N.stream_ptr(d256))
                else:
                    need = N.lib().hipt_hipt4k_workspace_bytes(pk256.ref, pk4k.ref, n, w_256, h_256, self.chunk)
                    ws = Fn.workspace(d256, need, slot)
                    N.call("hipt_hipt4k_forward", pk256.ref, pk4k.ref, N.ptr(region[lo:hi]), n, W, H, self.chunk, N.ptr(sub_cls),
                           N.ptr(out[lo:hi]), N.ptr(ws), ws.numel(), N.stream_ptr(d256))

            parts = self._parts(nreg)
            if parts == 1:
                launch(0, nreg, 0)
                return out, cls256
            cur = torch.cuda.current_stream(d256)
            key = (d256.index, parts)
            if key not in self._side_streams:
                self._side_streams[key] = [torch.cuda.Stream(device=d256) for _ in range(parts)]
            bounds = [nreg * k // parts for k in range(parts + 1)]
            for k, st in enumerate(self._side_streams[key]):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    launch(bounds[k], bounds[k + 1], 1 + k)
                for t in (region, out, cls256):
                    if t is not None:
                        t.record_stream(st)
            for st in self._side_streams[key]:
                cur.wait_stream(st)
            return out, cls256
        # two-device placement (hipt_4k.py:39-46): ViT-256 on device256, grid copied to device4k
        lay = N.ImageLayout(w_256, h_256, 256, 256, H, W * H, 3 * W * H)
        cls256 = self.model256.forward_features(region, layout=lay, nseq=nseq, chunk=self.chunk)
        tokens = cls256.to(d4k, non_blocking=True).view(nreg, per, -1)
        return self.model4k.forward_tokens(tokens, w_256, h_256), cls256

    def _run_patch_split(self, region, u8, hwc, nreg, w_256, h_256, W, H, pk256, dev, want_cls256, parts):
        """Fewer regions than streams (the reference's batch of ONE region, extract_features_fp.py:159-171): the PATCHES of the
        call are spread over the streams instead.  One region is 256 patches = 257 rows per CU: every kernel of a single
        stream ends in a ragged last round of tiles (2 x 128 rows + 1); two half-size streams fill each other's idle CUs.
        The input is brought to the compute dtype once, each stream runs ViT-256 over its range of patches into the shared
        [CLS] grid, ViT-4K follows on the caller's stream."""
        import ctypes as C
        per, nseq = w_256 * h_256, nreg * w_256 * h_256
        lay = N.ImageLayout(w_256, h_256, 256, 256, H, W * H, 3 * W * H)
        cur = torch.cuda.current_stream(dev)
        if u8:
            # raw RGB bytes: normalised once into fp32 (ToTensor + Normalize(0.5, 0.5), the bits torch computes), then the float path -- the
            # pixel-reading embedding below.  (Not the bf16 copy: the ranges must take the kernels the whole-region call takes -- embed32.hip hands
            # the first block its operands -- or a region's bits would depend on patch_streams.)
            buf = Fn.workspace(dev, nreg * 3 * W * H * 4, ("u8f32", cur.cuda_stream))[:nreg * 3 * W * H * 4].view(torch.float32).view(nreg, 3, W, H)
            N.call("hipt_u8_normalize", N.ptr(region), int(hwc), nreg, W * H, N.ptr(buf), N.HIPT_F32, N.stream_ptr(dev))
            region, u8, hwc = buf, False, False
        kind = 0
        # fp32 pixels and a patch embedding that reads them itself (csrc/embed32.hip): no copy in the compute dtype
        # (that kernel loads 16 bytes at a time: a region view that starts off a 16-byte boundary takes the converted copy instead)
        px = (not u8) and region.data_ptr() % 16 == 0 and N.lib().hipt_vit256_range_px_workspace_bytes(pk256.ref, C.byref(lay), 16, self.chunk) > 0
        nb = 0 if px else N.lib().hipt_image_compute_bytes(pk256.ref, C.byref(lay), nseq, kind)
        img = region
        if nb:
            img = Fn.workspace(dev, nb, ("img", cur.cuda_stream))
            N.call("hipt_image_to_compute", pk256.ref, N.ptr(region), kind, C.byref(lay), nseq, N.ptr(img), N.stream_ptr(dev))
        cls256 = torch.empty((nseq, pk256.w.dim), dtype=torch.float32, device=dev)
        bounds = [(nseq * k // parts) // 16 * 16 for k in range(parts)] + [nseq]  # whole 16-sequence groups: whole MFMA row fragments
        key = (dev.index, parts)
        if key not in self._side_streams:
            self._side_streams[key] = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        for k, st in enumerate(self._side_streams[key]):
            lo, n = bounds[k], bounds[k + 1] - bounds[k]
            if n <= 0:
                continue
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                if px:
                    need = N.lib().hipt_vit256_range_px_workspace_bytes(pk256.ref, C.byref(lay), n, self.chunk)
                else:
                    need = N.lib().hipt_vit256_range_workspace_bytes(pk256.ref, n, self.chunk)
                ws = Fn.workspace(dev, need, 1 + k)
                N.call("hipt_vit256_forward_range_px" if px else "hipt_vit256_forward_range", pk256.ref, N.ptr(img), C.byref(lay), lo, n,
                       self.chunk, N.ptr(cls256[lo:lo + n]), N.ptr(ws), ws.numel(), N.stream_ptr(dev))
            for t in (region, cls256):
                t.record_stream(st)
        for st in self._side_streams[key]:
            cur.wait_stream(st)
        out = self.model4k.forward_tokens(cls256.view(nreg, per, -1), w_256, h_256)
        return out, (cls256 if want_cls256 else None)

    def _parts(self, nreg: int) -> int:
        """Streams a call of `nreg` regions is spread over (at most `self.streams`).  tools/streams_by_regions_bench.py, regions/s by (R, S):
        R = 2: 270 / 278 (S = 1 / 2); R = 4: 300 / 298; R = 6: 310 / 308 / 284 (S = 3); R = 8: 314 / 309 / 291; R = 12: 308 / 316 / 303;
        R = 16: 314 / 319 / 312; R = 24: 313 / 318 / 317 (bench.py's step, with CLAM_SB between the calls: 324 / 325 / 328).  Two or three regions: one
        stream each pays (a region alone is two rounds of tiles + 2 tiles: the other stream fills the third round); from 4 to 11 regions ONE stream
        is best -- parts of 2-4 regions mostly pay each other's launch tails; from 12 on, a stream per six regions.  A region's bits do not
        depend on the partition (tests)."""
        smax = max(1, int(self.streams))
        if nreg <= 3:
            return min(smax, nreg, 2)
        if nreg < 12:
            return 1
        return max(1, min(smax, nreg // 6))

    def forward(self, x):
        """[R,3,W',H'] float -> [R,192] ViT-4K [CLS] features (hipt_4k.py:48-76; the reference takes R = 1).
        uint8 input ([R,3,W',H'] or interleaved [R,W',H',3]) is raw RGB: eval_transforms (ToTensor + Normalize(0.5, 0.5),
        hipt_model_utils.py:113-118) is applied on the device."""
        return self._run(x, want_cls256=False)[0]

    def _get_region_attention_scores(self, region, scale=1):
        """hipt_4k.py:121-164: the hierarchical attention maps of one region, the input of the heat-map code.

        ``region``: a ``PIL.Image`` / ``[W', H', 3] uint8`` array (goes through ``eval_transforms`` as in the reference) or an
        already normalised ``[1, 3, W', H']`` float tensor.  Returns ``(patches [n, 256/s, 256/s, 3] uint8 array,
        attention_256 [n, heads, 256/s, 256/s] array, attention_4k [heads, W/s, H/s] array)`` with ``n = w_256 * h_256`` patches.

        What the reference takes from ``get_last_selfattention`` is ``[:, :, 0, 1:]`` -- the [CLS] query's row -- so both maps come
        from the one-query kernels (``get_last_selfattention_cls``: ``hipt_vit_cls_attention``); the ``[n, 6, 257, 257]``
        probability tensor (406 MB for a 4096 x 4096 region) is never built."""
        import numpy as np

        from .hipt_model_utils import eval_transforms, tensorbatch2im
        if torch.is_tensor(region) and region.dim() == 4:
            x = region
        else:
            x = eval_transforms()(region).unsqueeze(dim=0)  # :135
        if x.shape[0] != 1:
            raise ValueError("_get_region_attention_scores describes ONE region, as in the reference (hipt_4k.py:138-146)")
        batch, w_256, h_256 = self.prepare_img_tensor(x)
        if w_256 == 0 or h_256 == 0:
            raise ValueError(f"region {tuple(x.shape)} is smaller than one 256x256 patch")
        d256, d4k = self.model256.weight_device, self.model4k.weight_device
        n = w_256 * h_256
        # 'b c p1 p2 w h -> (b p1 p2) c w h' (:137-139): patch k = p1 * h_256 + p2
        b256 = batch.to(d256).float().unfold(2, 256, 256).unfold(3, 256, 256).permute(0, 2, 3, 1, 4, 5).reshape(n, 3, 256, 256).contiguous()
        with torch.no_grad():
            features_cls256 = self.model256(b256)  # :141
            a256 = self.model256.get_last_selfattention_cls(b256)  # [n, heads, 257] = get_last_selfattention(b256)[:, :, 0, :]  (:143)
            nh = a256.shape[1]
            a256 = a256[:, :, 1:].reshape(n, nh, 16, 16)  # :145-146
            a256 = torch.nn.functional.interpolate(a256, scale_factor=int(16 / scale), mode="nearest").cpu().numpy()  # :147
            grid = features_cls256.reshape(w_256, h_256, -1).transpose(0, 1).transpose(0, 2).unsqueeze(dim=0).to(d4k)  # :149-150
            a4k = self.model4k.get_last_selfattention_cls(grid)  # [1, heads, 1 + n]  (:153)
            nh = a4k.shape[1]
            a4k = a4k[0, :, 1:].reshape(nh, w_256, h_256)  # :155-156
            a4k = torch.nn.functional.interpolate(a4k.unsqueeze(0), scale_factor=int(256 / scale), mode="nearest")[0].cpu().numpy()  # :157
            if scale != 1:
                b256 = torch.nn.functional.interpolate(b256, scale_factor=(1 / scale), mode="nearest")  # :159-160
        return tensorbatch2im(b256), np.asarray(a256), np.asarray(a4k)

    def forward_asset_dict(self, x: torch.Tensor):
        """hipt_4k.py:79-118: intermediate features as numpy arrays."""
        if x.shape[0] != 1:
            raise ValueError("forward_asset_dict describes ONE region ([1,3,W,H]), as in the reference (hipt_4k.py:96-97)")
        out, cls256 = self._run(x, want_cls256=True)
        f256 = cls256.detach().cpu()
        mean256 = f256.mean(dim=0).unsqueeze(dim=0)
        f4k = out.detach().cpu()
        return {
            'features_cls256': f256.numpy(),
            'features_mean256': mean256.numpy(),
            'features_cls4k': f4k.numpy(),
            'features_mean256_cls4k': torch.cat([mean256, f4k], dim=1).numpy(),
        }
