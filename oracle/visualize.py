"""ORACLE (test infrastructure, never imported by the product): NumPy restatement of the visualisation
lines of examples/predict_full_patched.py:88-110 -- class colours and the float64 overlay blend."""
import numpy as np


def colorize(pred: np.ndarray, id_colors: dict) -> np.ndarray:
    """predict_full_patched.py:89-95: zeros, then `colored[pred == id] = color` per class."""
    h, w = pred.shape[:2]
    colored = np.zeros((h, w, 3), dtype=np.uint8)
    for cid, color in id_colors.items():
        colored[pred == cid] = color
    return colored


def overlay(img: np.ndarray, colored: np.ndarray, alpha: float = 0.6) -> np.ndarray:
    """predict_full_patched.py:108-110."""
    return (img * alpha + colored * (1 - alpha)).astype(np.uint8)
