"""Host-side plumbing of the nodes (PIL / numpy / scipy), bit-exact with the reference's
utils/image_utils.py: size rounding (:12-49), float->uint8 frame conversion (:106-116), mask
conversion + diamond dilation (:126-175), outpaint canvas and border masks (:200-252), output
packing (:276-290).  Everything here is integer/byte work on small host arrays; the pixels then
go to the MI355X once, as uint8.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.ndimage
import torch
from PIL import Image


@dataclass
class ImageConfig:
    width: int
    height: int
    mask_dilates: int
    flow_mask_dilates: int
    input_size: tuple[int, int]
    video_length: int
    process_size: tuple[int, int] = field(init=False)

    def __post_init__(self) -> None:
        self.process_size = (self.width - self.width % 8, self.height - self.height % 8)


@dataclass
class ImageOutpaintConfig(ImageConfig):
    width_scale: float
    height_scale: float
    process_size: tuple[int, int] = field(init=False)
    outpaint_size: tuple[int, int] = field(init=False)

    def __post_init__(self) -> None:
        self.process_size = (self.width - self.width % 8, self.height - self.height % 8)
        pw, ph = int(self.width_scale * self.width), int(self.height_scale * self.height)
        self.outpaint_size = (pw - pw % 8, ph - ph % 8)


def image_to_uint8_frames(images: torch.Tensor) -> np.ndarray:
    """IMAGE [T,H,W,3] float in [0,1] -> uint8 [T,H,W,3] (truncating, like np.astype in :112)."""
    a = images.detach().cpu().numpy()
    return (a * 255).clip(0, 255).astype(np.uint8)


def resize_frames(frames: np.ndarray, size: tuple[int, int]) -> np.ndarray:
    """PIL default (bicubic) resize to (width, height) when the size differs (:98-103)."""
    if (frames.shape[2], frames.shape[1]) == tuple(size):
        return frames
    return np.stack([np.array(Image.fromarray(f).resize(size)) for f in frames], 0)


def _dilate(mask_u8: np.ndarray, iterations: int) -> np.ndarray:
    if iterations > 0:
        return scipy.ndimage.binary_dilation(mask_u8, iterations=iterations).astype(np.uint8)
    return (mask_u8 > 0.1).astype(np.uint8)


def read_masks(masks: torch.Tensor, config: ImageConfig) -> tuple[np.ndarray, np.ndarray]:
    """MASK [T|1,H,W] float -> (flow_masks, masks_dilated), each uint8 {0,1} [T,h,w] (:142-175)."""
    flow_masks, masks_dilated = [], []
    for m in masks:
        m = m.detach().cpu()
        if m.dtype == torch.float32:
            m = (m * 255).clamp(0, 255).byte()
        img = Image.fromarray(m.numpy())
        if config.process_size != config.input_size:
            img = img.resize(config.process_size)
        arr = np.array(img.convert("L"))
        flow_masks.append(_dilate(arr, config.flow_mask_dilates))
        masks_dilated.append(_dilate(arr, config.mask_dilates))
    if len(flow_masks) == 1:
        flow_masks = flow_masks * config.video_length
        masks_dilated = masks_dilated * config.video_length
    return np.stack(flow_masks, 0), np.stack(masks_dilated, 0)


def prepare_frames_and_masks(frames_u8: np.ndarray, mask: torch.Tensor, config: ImageConfig):
    """-> (frames uint8 [T,h,w,3], flow_masks uint8 [T,h,w], masks_dilated uint8 [T,h,w])."""
    frames = resize_frames(frames_u8, config.process_size)
    flow_masks, masks_dilated = read_masks(mask, config)
    return frames, flow_masks, masks_dilated


def outpaint_geometry(config: ImageOutpaintConfig):
    """-> ((pw, ph), (hs, ws), flow_mask u8 [ph,pw], mask u8 [ph,pw]): canvas size, frame offset and the two static
    border masks of extrapolation (:200-252)."""
    rw, rh = config.process_size
    pw, ph = config.outpaint_size
    ws, hs = int((pw - rw) / 2), int((ph - rh) / 2)
    dh = 4 if hs > 10 else 0
    dw = 4 if ws > 10 else 0
    mask = np.ones((ph, pw), dtype=np.uint8)
    mask[hs + dh:hs + rh - dh, ws + dw:ws + rw - dw] = 0
    flow_mask = mask.copy()
    mask[hs:hs + rh, ws:ws + rw] = 0
    return (pw, ph), (hs, ws), flow_mask, mask


def extrapolation(frames_u8: np.ndarray, config: ImageOutpaintConfig):
    """Outpaint canvas + static border masks (:200-252), host path."""
    frames = resize_frames(frames_u8, config.process_size)
    T, rh, rw, _ = frames.shape
    (pw, ph), (hs, ws), flow_mask, mask = outpaint_geometry(config)
    canvas = np.zeros((T, ph, pw, 3), dtype=np.uint8)
    canvas[:, hs:hs + rh, ws:ws + rw] = frames
    return canvas, np.repeat(flow_mask[None], T, 0), np.repeat(mask[None], T, 0)


def handle_output(composed_u8: np.ndarray | torch.Tensor, flow_masks: torch.Tensor, masks_dilated: torch.Tensor):
    """-> (IMAGE float32 [T,h,w,3] = k/255, FLOW_MASK [T,h,w] float, MASK_DILATE [T,h,w] float) (:276-290)."""
    comp = torch.as_tensor(composed_u8)
    images = torch.from_numpy(comp.cpu().numpy().astype(np.float32) / 255.0)
    return images, flow_masks.squeeze(), masks_dilated.squeeze()
