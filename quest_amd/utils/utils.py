"""Pool layout tags (reference: quest/utils/utils.py:1-5) + this build's row-rotated NHD."""


class TensorLayout:
    NHD = 0  # pool layer [pages, 2, page_size, heads, dim]
    HND = 1  # pool layer [pages, 2, heads, page_size, dim]
    # EXTENSION (include/quest_hip.h QUEST_LAYOUT_NHD_ROT): the NHD shape, but inside entry e's row of heads the K (max)
    # vector of head h sits in head slot h ^ (e & rot) and its V (min) vector in that slot ^ flip -- every head's 256-byte
    # pieces then cycle through all values of address bits 8-9, which MI355X does not serve equally fast under mixed
    # traffic.  Same results bit for bit; a pool in this layout must be written and read through this library
    # (`to_logical` below gives the NHD view back for inspection).
    NHD_ROT = 2

    FORMAT2STR = {0: "NHD", 1: "HND", 2: "NHD_ROT"}

    @staticmethod
    def parse(layout) -> int:
        if isinstance(layout, str):
            if layout not in ("NHD", "HND", "NHD_ROT"):
                raise KeyError("Invalide kv_layout {}".format(layout))
            return getattr(TensorLayout, layout)
        if layout not in (0, 1, 2):
            raise KeyError("Invalide kv_layout {}".format(layout))
        return int(layout)

    @staticmethod
    def rotation(num_heads: int):
        """(rot, flip) of the NHD_ROT layout for a pool of ``num_heads`` heads (csrc/quest_common.cuh pool_strides)."""
        low = num_heads & -num_heads
        return min(low, 4) - 1, (min(low, 32) - 1) & ~3

    @staticmethod
    def to_logical(pages, layout: int):
        """Pages ``[..., 2, S, H, D]`` (NHD / NHD_ROT) or ``[..., 2, H, S, D]`` (HND) of a pool layer -> the NHD view
        ``[..., 2, S, H, D]`` with head h at index h (a copy for the other two layouts).  Inspection / test aid: the
        product path never permutes a pool."""
        import torch

        layout = TensorLayout.parse(layout)
        if layout == TensorLayout.HND:
            return pages.transpose(-3, -2)
        if layout == TensorLayout.NHD:
            return pages
        S, H = pages.shape[-3], pages.shape[-2]
        rot, flip = TensorLayout.rotation(H)
        e = torch.arange(S, device=pages.device).view(S, 1)
        h = torch.arange(H, device=pages.device).view(1, H)
        k_slot = h ^ (e & rot)                      # [S, H]
        slot = torch.stack([k_slot, k_slot ^ flip])  # [2, S, H]
        idx = slot.view((1,) * (pages.dim() - 4) + (2, S, H, 1)).expand(pages.shape)
        return torch.gather(pages, -2, idx)
