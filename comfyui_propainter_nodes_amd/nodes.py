"""ComfyUI node API -- the drop-in boundary.  INPUT_TYPES / RETURN_TYPES / RETURN_NAMES / FUNCTION /
CATEGORY, method names, keyword names, error behaviour and the node mappings are those of the
reference's propainter_nodes.py (:21-35, :38-154, :157-310, :313-321); the work behind them runs
on the MI355X through libpropainter_mi355.
"""
from __future__ import annotations

import torch

from .image_utils import (
    ImageConfig,
    ImageOutpaintConfig,
    extrapolation,
    handle_output,
    image_to_uint8_frames,
    prepare_frames_and_masks,
)
from .pipeline import ProPainterConfig, initialize_models, run_inpainting


def get_torch_device() -> torch.device:
    """comfy.model_management.get_torch_device() inside ComfyUI (propainter_nodes.py:109,247);
    outside ComfyUI (bench, tests) the first visible MI355X.  There is no CPU execution path."""
    try:
        from comfy import model_management  # type: ignore

        return model_management.get_torch_device()
    except ImportError:
        if not torch.cuda.is_available():
            raise RuntimeError("ProPainter (MI355X build) needs a ROCm GPU; none is visible") from None
        return torch.device("cuda", torch.cuda.current_device())


def check_inputs(frames: torch.Tensor, masks: torch.Tensor) -> Exception | None:
    if frames.size(dim=0) <= 1:
        raise Exception(f"""Image length must be greater than 1, but got:
                        Image length: ({frames.size(dim=0)})""")
    if frames.size(dim=0) != masks.size(dim=0) and masks.size(dim=0) != 1:
        raise Exception(f"""Image and Mask must have the same length or Mask have length 1, but got:
                        Image length: {frames.size(dim=0)}
                        Mask length: {masks.size(dim=0)}""")
    if frames.size(dim=1) != masks.size(dim=1) or frames.size(dim=2) != masks.size(dim=2):
        raise Exception(f"""Image and Mask must have the same dimensions, but got:
                        Image: ({frames.size(dim=1)}, {frames.size(dim=2)})
                        Mask: ({masks.size(dim=1)}, {masks.size(dim=2)})""")


_COMMON_INPUTS = {
    "mask_dilates": ("INT", {"default": 5, "min": 0, "max": 100}),
    "flow_mask_dilates": ("INT", {"default": 8, "min": 0, "max": 100}),
    "ref_stride": ("INT", {"default": 10, "min": 1, "max": 100}),
    "neighbor_length": ("INT", {"default": 10, "min": 2, "max": 300}),
    "subvideo_length": ("INT", {"default": 80, "min": 1, "max": 300}),
    "raft_iter": ("INT", {"default": 20, "min": 1, "max": 100}),
    "fp16": (["enable", "disable"],),
}


def _finish(models, frames_u8, flow_masks, masks_dilated, config):
    composed = run_inpainting(models, frames_u8, flow_masks, masks_dilated, config)
    dev = config.device
    fm = torch.from_numpy(flow_masks).float().to(dev)
    md = torch.from_numpy(masks_dilated).float().to(dev)
    return handle_output(composed, fm, md)


class ProPainterInpaint:
    """ComfyUI Node for performing inpainting on video frames using ProPainter."""

    def __init__(self):
        pass

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "image": ("IMAGE",),
                "mask": ("MASK",),
                "width": ("INT", {"default": 640, "min": 0, "max": 2560}),
                "height": ("INT", {"default": 360, "min": 0, "max": 2560}),
                **_COMMON_INPUTS,
            },
        }

    RETURN_TYPES = ("IMAGE", "MASK", "MASK")
    RETURN_NAMES = ("IMAGE", "FLOW_MASK", "MASK_DILATE")
    FUNCTION = "propainter_inpainting"
    CATEGORY = "ProPainter"

    def propainter_inpainting(self, image: torch.Tensor, mask: torch.Tensor, width: int, height: int, mask_dilates: int,
                              flow_mask_dilates: int, ref_stride: int, neighbor_length: int, subvideo_length: int,
                              raft_iter: int, fp16: str) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        check_inputs(image, mask)
        device = get_torch_device()
        frames_u8 = image_to_uint8_frames(image)
        video_length = image.size(dim=0)
        input_size = (frames_u8.shape[2], frames_u8.shape[1])
        image_config = ImageConfig(width, height, mask_dilates, flow_mask_dilates, input_size, video_length)
        config = ProPainterConfig(ref_stride, neighbor_length, subvideo_length, raft_iter, fp16, video_length, device,
                                  image_config.process_size)
        frames_u8, flow_masks, masks_dilated = prepare_frames_and_masks(frames_u8, mask, image_config)
        models = initialize_models(device, config.fp16)
        print(f"\nProcessing  {config.video_length} frames...")
        return _finish(models, frames_u8, flow_masks, masks_dilated, config)


class ProPainterOutpaint:
    """ComfyUI Node for performing outpainting on video frames using ProPainter."""

    def __init__(self):
        pass

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "image": ("IMAGE",),
                "width": ("INT", {"default": 640, "min": 0, "max": 2560}),
                "height": ("INT", {"default": 360, "min": 0, "max": 2560}),
                "width_scale": ("FLOAT", {"default": 1.2, "min": 0.0, "max": 10.0, "step": 0.01}),
                "height_scale": ("FLOAT", {"default": 1.0, "min": 0.0, "max": 10.0, "step": 0.01}),
                **_COMMON_INPUTS,
            },
        }

    RETURN_TYPES = ("IMAGE", "MASK", "INT", "INT")
    RETURN_NAMES = ("IMAGE", "OUTPAINT_MASK", "output_width", "output_height")
    FUNCTION = "propainter_outpainting"
    CATEGORY = "ProPainter"

    def propainter_outpainting(self, image: torch.Tensor, width: int, height: int, width_scale: float, height_scale: float,
                               mask_dilates: int, flow_mask_dilates: int, ref_stride: int, neighbor_length: int,
                               subvideo_length: int, raft_iter: int, fp16: str) -> tuple[torch.Tensor, torch.Tensor, int, int]:
        device = get_torch_device()
        frames_u8 = image_to_uint8_frames(image)
        video_length = image.size(dim=0)
        input_size = (frames_u8.shape[2], frames_u8.shape[1])
        image_config = ImageOutpaintConfig(width, height, mask_dilates, flow_mask_dilates, input_size, video_length,
                                           width_scale, height_scale)
        config = ProPainterConfig(ref_stride, neighbor_length, subvideo_length, raft_iter, fp16, video_length, device,
                                  image_config.outpaint_size)
        frames_u8, flow_masks, masks_dilated = extrapolation(frames_u8, image_config)
        models = initialize_models(device, config.fp16)
        print(f"\nProcessing  {config.video_length} frames...")
        output_frames, output_masks, _ = _finish(models, frames_u8, flow_masks, masks_dilated, config)
        output_width, output_height = config.process_size
        return output_frames, output_masks, output_width, output_height


NODE_CLASS_MAPPINGS = {
    "ProPainterInpaint": ProPainterInpaint,
    "ProPainterOutpaint": ProPainterOutpaint,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "ProPainterInpaint": "ProPainter Inpainting",
    "ProPainterOutpaint": "ProPainter Outpainting",
}
