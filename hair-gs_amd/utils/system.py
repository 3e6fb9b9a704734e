"""Small filesystem helpers of the reference's utils/system.py."""
import os


def mkdir_p(folder_path):
    os.makedirs(folder_path, exist_ok=True)


def search_for_max_interation(folder):   # (sic: the reference's spelling)
    return max(int(name.split("_")[-1]) for name in os.listdir(folder))
