"""ComfyUI custom-node entry point (drop this repository into ComfyUI/custom_nodes/)."""
if __package__:
    # loaded by ComfyUI as a package: any failure inside the import (missing library, scipy, PIL ...) must surface as is
    from .comfyui_propainter_nodes_amd.nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS
else:  # imported as a top-level module (tests, bench)
    from comfyui_propainter_nodes_amd.nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
