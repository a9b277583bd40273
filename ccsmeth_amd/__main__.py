"""python -m ccsmeth_amd call_mods ... | call_freqb ... | trainm ... | train ... | extract ...   (the sub-commands of the reference CLI on / next to the hot path)"""
import sys


def main():
    cmds = ("call_mods", "call_freqb", "trainm", "train", "extract")
    if len(sys.argv) < 2 or sys.argv[1] not in cmds:
        sys.exit("usage: python -m ccsmeth_amd call_mods -i in.bam -m model.ckpt -o out_prefix [options]\n"
                 "       python -m ccsmeth_amd call_freqb --input_bam aligned.modbam.bam --ref genome.fa -o out_prefix [options]\n"
                 "       python -m ccsmeth_amd trainm --train_file f.tsv --valid_file v.tsv --model_dir dir [options]\n"
                 "       python -m ccsmeth_amd extract -i hifi.bam -o features.tsv [options]")
    if sys.argv[1] == "call_mods":
        from .call_mods import main as cm
        cm(sys.argv[2:])
    elif sys.argv[1] == "call_freqb":
        from .call_mods_freq_bam import main as cf
        cf(sys.argv[2:])
    elif sys.argv[1] == "trainm":
        from .trainm import main as tm
        tm(sys.argv[2:])
    elif sys.argv[1] == "train":
        from .trainm import main_train as tr
        tr(sys.argv[2:])
    else:
        from .extract_cli import main as em
        em(sys.argv[2:])


if __name__ == "__main__":
    main()
