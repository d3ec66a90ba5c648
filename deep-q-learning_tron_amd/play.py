"""Batched evaluation — the rating sweep of the reference's play.py (play.py:72-98) on VecTron.

The reference plays 13 x 10 000 "fair"/"ice" games one after another, sweeping `slide_pram`
from 0 to 0.36 in steps of 0.03, and prints player 1's win rate per value.  Here every game of
every slide value is one env of a single batch (the slide probability is a per-env quantity on
the device), stepped until all are finished.  The pygame menu / window of play.py:22-45,100-107
is out of scope."""
import argparse

import torch

from tron.vec import VecTron, pop_up_planes


def model_actions(model, planes, prob_plane, env_vec):
    """model.act on a batch, in the two calling conventions of Game.main_loop (game.py:296-304)."""
    with torch.no_grad():
        if getattr(model, "wants_prob_plane", False):
            a = model.act(torch.cat([planes, prob_plane], 1))
        else:
            a = model.act(planes, env_vec)
    return torch.as_tensor(a).reshape(-1).to(torch.int8)


def rating(model, model2=None, n_games=10000, slides=None, width=10, gamemode="ice", fair=True, seed=0x5EED,
           max_steps=None, verbose=True):
    """Returns a list of dicts {slide, p1_win, p2_win, draw, p1_rate} (play.py:76-98).
    model2="minimax" seats MinimaxPlayer(2, "voronoi") as player 2 — the "minimax rating" that
    ACKTR.py:408-421 logs (there it is played net against net)."""
    model2 = model2 or model
    slides = [0.03 * i for i in range(13)] if slides is None else list(slides)     # play.py:74,98
    n = n_games * len(slides)
    S = width + 2
    env = VecTron(n, width, mode=gamemode, fair=fair, seed=seed, obs_format="codes", reward="ddqn")
    slide_t = torch.tensor(slides, dtype=torch.float64, device=env.device).repeat_interleave(n_games)
    env.set_slide(slide_t)
    obs = env.reset()
    st = env.state()
    # what main_loop feeds beside the planes: get_multy(0) to player 1, [get_rate()] to player 2
    degree = st["degree"].to(torch.float32)
    env1 = torch.stack([degree, st["weight"][:, 0].to(torch.float32)], 1)
    env2 = -((degree - 30) * 0.6) / 100           # [n]: main_loop hands player 2 torch.tensor([get_rate()]) per game
    prob_plane = ((-slide_t * 100) * (10 / 6) + 30).to(torch.float32).view(n, 1, 1, 1).expand(n, 1, S, S)
    for _ in range(max_steps or width * width):
        planes = pop_up_planes(obs.reshape(2 * n, S, S)).view(n, 2, 3, S, S)
        a1 = model_actions(model, planes[:, 0], prob_plane, env1)
        a2 = env.minimax_actions(2) if isinstance(model2, str) else model_actions(model2, planes[:, 1], prob_plane, env2)
        obs, _, done, _ = env.step(torch.stack([a1, a2], 1), autoreset=False)
        if bool(done.all()):
            break
    winner = env.state()["winner"].view(len(slides), n_games)
    out = []
    for i, s in enumerate(slides):
        p1, p2 = int((winner[i] == 1).sum()), int((winner[i] == 2).sum())
        out.append(dict(slide=s, p1_win=p1, p2_win=p2, draw=n_games - p1 - p2, p1_rate=p1 / max(p1 + p2, 1)))
        if verbose:
            print("Player 1:{} \nPlayer 2:{}\np1's win rating {}\nprob={}".format(p1, p2, out[-1]["p1_rate"], s))
    env.close()
    return out


ARCHS = ("dqn", "map", "test", "net3", "net4", "mul")


def load_player(path=None, arch="dqn", width=None, device=None):
    """A player network from a `.bak` state_dict, the way play.py:53-61 builds its two: the reference wraps the net
    in Brain(net, args, acktr=True) first — KFACOptimizer splits every bias into an AddBias layer (kfac.py:80-96,145) —
    so an ACKTR checkpoint's keys read `conv1.module.weight` / `conv1.add_bias._bias`; A2C / DQN checkpoints keep
    `conv1.weight` / `conv1.bias`.  Both layouts are recognised from the keys.  arch: "map" = MapNet
    (ACKTR_player3map_*.bak), "test" = TestNet (ACKTR_player2make_dyna_model.bak), "net3" / "net4" / "mul" the other
    ACNet classes, "dqn" = Net/DQNNet.Net (DDQN.bak).  Files are read with weights_only=True."""
    from config import MAP_WIDTH
    import Net.ACNet as A
    from Net.DQNNet import Net as DQNNet
    from Net.kfac import split_biases
    width = MAP_WIDTH if width is None else width
    device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
    sd = None
    if path:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "local" in sd and "target" in sd:      # DDQN.save_checkpoint's full file
            sd = sd["target"]
    if arch == "dqn":
        in_ch = 3 if sd is None else int(sd["conv1.module.weight" if "conv1.module.weight" in sd else "conv1.weight"].shape[1])
        net = DQNNet(in_ch, width)
    else:
        net = {"map": A.MapNet, "test": A.TestNet, "net3": A.Net3, "net4": A.Net4, "mul": A.Mulnet}[arch](width)
    if sd is not None:
        if any(k.endswith(".add_bias._bias") for k in sd):                 # the checkpoint was saved under K-FAC
            if arch == "dqn":
                # DQNNet.Net runs its layers through the HIP operators, which read conv.weight / conv.bias: fold the split
                # keys back (`x.module.weight` -> `x.weight`, `x.add_bias._bias` [C, 1] -> `x.bias` [C]) instead of wrapping
                sd = {k.replace(".module.", ".").replace(".add_bias._bias", ".bias"):
                      (v.reshape(-1) if k.endswith(".add_bias._bias") else v) for k, v in sd.items()}
            else:
                split_biases(net)
        net.load_state_dict(sd)
    return net.to(device).eval()


def main(args):
    nets = [load_player(args.p1, args.arch1), load_player(args.p2, args.arch2)]
    rating(nets[0], nets[1], n_games=args.games)


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--p1', required=False, help="state_dict (.bak) of player 1's net")
    parser.add_argument('--p2', required=False, help="state_dict (.bak) of player 2's net")
    parser.add_argument('--arch1', default="map", choices=ARCHS, help="play.py:53 seats a MapNet as player 1")
    parser.add_argument('--arch2', default="test", choices=ARCHS, help="play.py:58 seats a TestNet as player 2")
    parser.add_argument('--games', type=int, default=10000)
    main(parser.parse_args())
