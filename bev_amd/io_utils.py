"""Tiny text readers the calibration loaders need (KITTI-style "name: v v v" files and plain
number tables); mirrors read_txt_to_dict / read_txt_to_array of /root/reference/bev/io/utils.py:7-41.
Video / label I/O of the reference is host codec work outside the warp path."""
import numpy as np

_NUMERIC_CHARS = set("0123456789.e+- ")


def read_txt_to_dict(path):
    """{"name": ndarray | str} from lines of "name: values"; values that parse as floats become arrays."""
    data = {}
    with open(path, "r") as f:
        for line in f:
            if len(line) <= 1:
                continue
            key, value = line.split(":", 1)
            value = value.strip()
            data[key] = value
            if _NUMERIC_CHARS.issuperset(value):
                try:
                    data[key] = np.array([float(tok) for tok in value.split(" ")])
                except ValueError:
                    pass
    return data


def read_txt_to_array(path):
    with open(path) as f:
        return np.array([[float(tok) for tok in line.split()] for line in f.readlines()])
