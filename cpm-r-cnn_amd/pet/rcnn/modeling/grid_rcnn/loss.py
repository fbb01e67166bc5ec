"""calc_sub_regions (counterpart of pet/rcnn/modeling/grid_rcnn/loss.py:244-273): the 28x28 window of the 56x56
heat map that each grid point predicts (Grid R-CNN Plus)."""


def calc_sub_regions(grid_points, grid_size, whole_map_size):
    half = whole_map_size // 4 * 2

    def start(idx):
        if idx == 0:
            return 0
        if idx == grid_size - 1:
            return half
        return max(int((idx / (grid_size - 1) - 0.25) * whole_map_size), 0)

    out = []
    for i in range(grid_points):
        sx, sy = start(i // grid_size), start(i % grid_size)
        out.append((sx, sy, sx + half, sy + half))
    return out
