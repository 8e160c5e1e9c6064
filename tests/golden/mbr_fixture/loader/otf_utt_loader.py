"""TEST FIXTURE standing in for `loader.otf_utt_loader` when the UNCHANGED MBR training script
(trainer/train_transducer_mbr_bmuf_otfaug.py) is run for tests/golden/mbr_script_grads.npz: same three entry points
(`register`, `get_inputdim`, `dataloader`, loader/otf_utt_loader.py:61-163), but the batches are seeded synthetic
features instead of audio (the reference loader needs PyKaldi for its filter banks).  Used identically for the run on
the reference's own modules (golden) and the run on the drop-in packages (test)."""
import torch


def register(parser):
    a = parser.add_argument
    a('--lctx', type=int, default=1)
    a('--rctx', type=int, default=1)
    a('--feats_dim', type=int, default=80)
    a('--batch_size', type=int, default=3)
    a('--fixture_seed', type=int, default=31)
    a('--fixture_batches', type=int, default=1)


def get_inputdim(args):
    return args.feats_dim * (args.lctx + 1 + args.rctx)


def dataloader(data_lst, rir, noise, args):
    g = torch.Generator().manual_seed(args.fixture_seed)
    B, D, V = args.batch_size, get_inputdim(args), args.output_dim
    for _ in range(args.fixture_batches):
        lens = torch.tensor([150, 142, 131][:B] + [130] * max(0, B - 3), dtype=torch.int32)   # sorted, as pack() needs
        data = torch.randn(B, int(lens.max()), D, generator=g)
        ali_lens = torch.tensor([5, 3, 4][:B] + [2] * max(0, B - 3), dtype=torch.int32)
        target = torch.full((B, int(ali_lens.max())), args.padding_idx, dtype=torch.int32)
        for b in range(B):
            target[b, :ali_lens[b]] = torch.randint(1, V, (int(ali_lens[b]),), generator=g, dtype=torch.int32)
        yield data, target, lens, ali_lens
