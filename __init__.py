"""ComfyUI custom-node entry point (drop this repository into ComfyUI/custom_nodes/)."""
try:
    from .comfyui_propainter_nodes_amd.nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS
except ImportError:  # imported as a top-level module (tests, bench)
    from comfyui_propainter_nodes_amd.nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
