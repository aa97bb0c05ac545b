"""Pool layout tags (reference: quest/utils/utils.py:1-5)."""


class TensorLayout:
    NHD = 0  # pool layer [pages, 2, page_size, heads, dim]
    HND = 1  # pool layer [pages, 2, heads, page_size, dim]

    FORMAT2STR = {0: "NHD", 1: "HND"}

    @staticmethod
    def parse(layout) -> int:
        if isinstance(layout, str):
            if not hasattr(TensorLayout, layout) or layout not in ("NHD", "HND"):
                raise KeyError("Invalide kv_layout {}".format(layout))
            return getattr(TensorLayout, layout)
        if layout not in (0, 1):
            raise KeyError("Invalide kv_layout {}".format(layout))
        return int(layout)
