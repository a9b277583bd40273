"""python -m ccsmeth_amd call_mods ...   (the sub-command of the reference CLI that sits on the hot path)"""
import sys


def main():
    if len(sys.argv) < 2 or sys.argv[1] != "call_mods":
        sys.exit("usage: python -m ccsmeth_amd call_mods -i in.bam -m model.ckpt -o out_prefix [options]")
    from .call_mods import main as cm
    cm(sys.argv[2:])


if __name__ == "__main__":
    main()
