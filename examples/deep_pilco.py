#!/usr/bin/env python
"""Deep-PILCO on the MI355X path: the loop of the reference's examples/deep_pilco_mm.py and
examples/deep_pilco_no_mm.py (collect experience -> fit the BNN dynamics -> MC-PILCO policy
search), written against prob_mbrl_amd.  Same stages, same calls, same checkpoint files
(`experience.pth.tar`, `latest_dynamics.pth.tar`, `latest_policy.pth.tar`); the environment is
the self-contained cart-pole of prob_mbrl_amd.envs (gym / Box2D are not part of this build).

    python examples/deep_pilco.py --ps_iters 5                 # moment matching (deep_pilco_mm)
    python examples/deep_pilco.py --no_mm --pol_batch_size 100 # particles only (deep_pilco_no_mm)
"""
import argparse
import copy
import datetime
import os
import sys
from functools import partial

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prob_mbrl_amd import algorithms, envs, models, utils  # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser('Deep-PILCO (MI355X build)')
    ap.add_argument('-e', '--env', type=str, default='Cartpole')
    ap.add_argument('-o', '--output_folder', type=str, default='~/.prob_mbrl_amd/')
    ap.add_argument('-s', '--seed', type=int, default=1)
    ap.add_argument('--n_initial_epi', type=int, default=1)
    ap.add_argument('--load_from', type=str, default=None)
    ap.add_argument('--pred_H', type=int, default=15)
    ap.add_argument('--control_H', type=int, default=40)
    ap.add_argument('--discount_factor', type=str, default=None)
    ap.add_argument('--prioritized_replay', action='store_true')
    ap.add_argument('--mm_groups', type=int, default=None)
    ap.add_argument('--no_mm', action='store_true', help='particles only (deep_pilco_no_mm.py)')
    ap.add_argument('--dyn_lr', type=float, default=1e-4)
    ap.add_argument('--dyn_opt_iters', type=int, default=2000)
    ap.add_argument('--dyn_batch_size', type=int, default=100)
    ap.add_argument('--dyn_drop_rate', type=float, default=0.1)
    ap.add_argument('--dyn_components', type=int, default=1,
                    help='> 1: GaussianMixtureDensity dynamics head (examples/deep_pilco_mm.py:117-121)')
    ap.add_argument('--dyn_shape', type=lambda s: [int(v) for v in s.split(',')], default=[200, 200])
    ap.add_argument('--pol_lr', type=float, default=1e-3)
    ap.add_argument('--pol_clip', type=float, default=1.0)
    ap.add_argument('--pol_drop_rate', type=float, default=0.1)
    ap.add_argument('--pol_opt_iters', type=int, default=1000)
    ap.add_argument('--pol_batch_size', type=int, default=100)
    ap.add_argument('--ps_iters', type=int, default=100)
    ap.add_argument('--pol_shape', type=lambda s: [int(v) for v in s.split(',')], default=[200, 200])
    ap.add_argument('--stop_when_done', action='store_true')
    ap.add_argument('--expl_noise', type=float, default=0.0)
    ap.add_argument('--resampling_period', type=int, default=499)
    ap.add_argument('--device', type=str, default='cuda:0')
    return ap.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    dev = torch.device(args.device)
    env = getattr(envs, args.env)()
    results_folder = os.path.join(os.path.expanduser(args.output_folder), 'mc_pilco', args.env,
                                  datetime.datetime.now().strftime('%Y_%m_%d_%H_%M_%S.%f'))
    os.makedirs(results_folder, exist_ok=True)
    results_filename = os.path.join(results_folder, 'experience.pth.tar')
    torch.save(args, os.path.join(results_folder, 'args.pth.tar'))

    D = env.observation_space.shape[0]
    U = env.action_space.shape[0]
    maxU, minU = env.action_space.high, env.action_space.low
    if args.discount_factor is not None:
        args.discount_factor = ((1.0 / args.control_H)**(2.0 / args.control_H)
                                if args.discount_factor == 'auto' else float(args.discount_factor))

    if args.dyn_components > 1:     # mixture-of-Gaussians head: (2 D + 1) n + 1 outputs
        density, dynE = models.GaussianMixtureDensity(D, args.dyn_components), (2 * D + 1) * args.dyn_components + 1
    else:
        density, dynE = models.DiagGaussianDensity(D), 2 * D
    dyn = models.DynamicsModel(
        models.mlp(D + U, dynE, args.dyn_shape,
                   dropout_layers=[models.CDropout(args.dyn_drop_rate * np.ones(h)) if args.dyn_drop_rate > 0
                                   else None for h in args.dyn_shape], nonlin=torch.nn.ReLU),
        reward_func=env.reward_func, output_density=density).float()
    pol = models.Policy(
        models.mlp(D, 2 * U, args.pol_shape,
                   dropout_layers=[models.BDropout(args.pol_drop_rate) if args.pol_drop_rate > 0 else None
                                   for h in args.pol_shape], nonlin=torch.nn.ReLU,
                   output_nonlin=partial(models.DiagGaussianDensity, U)), maxU, minU).float()
    exp = utils.ExperienceDataset()
    if args.load_from is not None:
        utils.load_checkpoint(args.load_from, dyn, pol, exp)
    opt1 = torch.optim.Adam(dyn.parameters(), args.dyn_lr)
    opt2 = torch.optim.Adam(pol.parameters(), args.pol_lr)
    dyn, pol = dyn.to(dev), pol.to(dev)

    env.seed(args.seed)
    rnd = lambda x, t: env.action_space.sample()  # noqa: E731
    initial_experience = args.control_H * args.n_initial_epi
    while exp.n_samples() < initial_experience:
        ret = utils.apply_controller(env, rnd, min(args.control_H, initial_experience - exp.n_samples() + 1),
                                     stop_when_done=args.stop_when_done)
        exp.append_episode(*ret, policy_params=[])
    if initial_experience > 0:
        exp.policy_parameters[-1] = copy.deepcopy(pol.state_dict())
    exp.save(results_filename)

    expl_pol = lambda x, t: (pol(x) + args.expl_noise * rnd(x, t)).clip(minU, maxU)  # noqa: E731
    history = []
    for ps_it in range(args.ps_iters):
        # apply the policy
        new_exp = exp.n_samples() + args.control_H
        while exp.n_samples() < new_exp:
            ret = utils.apply_controller(env, expl_pol, min(args.control_H, new_exp - exp.n_samples() + 1),
                                         stop_when_done=args.stop_when_done)
            exp.append_episode(*ret, policy_params=[])
        exp.policy_parameters[-1] = copy.deepcopy(pol.state_dict())
        exp.save(results_filename)

        # fit the dynamics model
        X, Y = exp.get_dynmodel_dataset(deltas=True, return_costs=False)
        dyn.set_dataset(X.to(dev, torch.float32), Y.to(dev, torch.float32))
        utils.train_regressor(dyn, args.dyn_opt_iters, args.dyn_batch_size, True, opt1,
                              log_likelihood=dyn.output_density.log_prob)
        torch.save(dyn.state_dict(), os.path.join(results_folder, 'latest_dynamics.pth.tar'))

        # policy search
        x0 = exp.sample_states(args.pol_batch_size, timestep=0).to(dev, torch.float32).detach()
        losses = []
        print('Policy search iteration %d' % (ps_it + 1))
        algorithms.mc_pilco(x0, dyn, pol, args.pred_H, opt2, exp, args.pol_opt_iters,
                            discount=args.discount_factor, pegasus=True, mm_states=not args.no_mm,
                            mm_rewards=not args.no_mm, mm_groups=args.mm_groups, maximize=True,
                            clip_grad=args.pol_clip, resampling_period=args.resampling_period,
                            step_idx_to_sample=0, init_state_noise=1e-2 * x0.std(0),
                            prioritized_replay=args.prioritized_replay,
                            on_iteration=lambda i, loss, *a: losses.append(float(loss)))
        torch.save(pol.state_dict(), os.path.join(results_folder, 'latest_policy.pth.tar'))
        history.append(dict(episode_reward=float(np.sum(ret[2])), n_samples=exp.n_samples(),
                            first_loss=losses[0] if losses else None,
                            last_loss=losses[-1] if losses else None))
        print('  episode reward %.3f, predicted return %s -> %s' %
              (history[-1]['episode_reward'], history[-1]['first_loss'], history[-1]['last_loss']))
    return results_folder, history


if __name__ == '__main__':
    main()
