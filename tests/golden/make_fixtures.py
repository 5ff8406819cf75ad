"""Copies the data fixtures that the tests and bench.py need out of the reference tree.

The reference holds no tests or golden vectors, only data files. These are data, not source:
the Cornell box mesh, its twelve textures, every .lights file and every .rt_save quicksave.
Run once in the authoring container (needs /root/reference); the GPU box only sees the copies.
"""
import os
import shutil
import sys

REFERENCE = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def copy(src, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)


def main():
    data = os.path.join(REFERENCE, "data")
    out = os.path.join(HERE, "data")
    copy(os.path.join(data, "cornell_box.vks"), os.path.join(out, "cornell_box.vks"))
    for name in sorted(os.listdir(os.path.join(data, "cornell_box_textures"))):
        copy(os.path.join(data, "cornell_box_textures", name), os.path.join(out, "cornell_box_textures", name))
    for name in sorted(os.listdir(data)):
        if name.endswith(".lights"):
            copy(os.path.join(data, name), os.path.join(out, name))
    for root, _, files in os.walk(os.path.join(data, "saves")):
        for name in sorted(files):
            if name.endswith(".rt_save"):
                source = os.path.join(root, name)
                copy(source, os.path.join(out, os.path.relpath(source, data)))


if __name__ == "__main__":
    main()
