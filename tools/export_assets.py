"""Export the *data* assets of the reference that the hot path needs at run time.

Runs only in the build container (needs /root/reference).  It copies DATA, never source:
  * marker -> SMPL-X vertex ids        (loader/SSM2.json, loader/SSM2_withhand.json)
  * heel / toe vertex ids              (body_segments/{L,R}_Leg.json + foot_verts_id/*.npy,
                                        resolved exactly as opt_amass_temp.py:99-113 does, i.e.
                                        through CPython ``list(set(...))`` order -- SURVEY G6)
  * smoothness / infill statistics     (preprocess_stats/*.npz)
  * smoothness encoder weights         (runs/15217/Enc_last_model.pkl -> plain npz)
  * one example clip of per-frame fit results (res_opt_amass_perframe/TotalCapture clip 0),
    used as a realistic *input* distribution for tests / bench.
Output: lemo_amd/assets/*.npz (committed; the GPU box has no /root/reference).
"""
import json
import os
import sys

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lemo_amd', 'assets')


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(f'{REF}/loader/SSM2.json') as f:
        ids67 = list(json.load(f)['markersets'][0]['indices'].values())
    with open(f'{REF}/loader/SSM2_withhand.json') as f:
        ids81 = list(json.load(f)['markersets'][0]['indices'].values())
    foot = {}
    for side, seg in (('left', 'L_Leg'), ('right', 'R_Leg')):
        with open(f'{REF}/body_segments/{seg}.json') as f:
            # opt_amass_temp.py:99-113 -- order is CPython set iteration order, NOT sorted.
            verts = np.asarray(list(set(json.load(f)['verts_ind'])))
        for part in ('heel', 'toe'):
            m = np.load(f'{REF}/foot_verts_id/{side}_{part}_verts_id.npy')
            foot[f'{side}_{part}'] = verts[m].astype(np.int32)
    np.savez(os.path.join(OUT, 'vertex_ids.npz'),
             markers67=np.asarray(ids67, np.int32), markers81=np.asarray(ids81, np.int32), **foot)

    s = np.load(f'{REF}/preprocess_stats/preprocess_stats_smooth_withHand_global_markers.npz')
    np.savez(os.path.join(OUT, 'stats_smooth.npz'), **{k: s[k] for k in s.files})
    s = np.load(f'{REF}/preprocess_stats/preprocess_stats_infill_local_markers_4chan.npz')
    np.savez(os.path.join(OUT, 'stats_infill.npz'), **{k: s[k] for k in s.files})

    w = torch.load(f'{REF}/runs/15217/Enc_last_model.pkl', map_location='cpu')
    np.savez(os.path.join(OUT, 'smooth_enc_15217.npz'), **{k: v.numpy() for k, v in w.items()})

    d = f'{REF}/res_opt_amass_perframe/TotalCapture'
    np.savez(os.path.join(OUT, 'example_clip0.npz'),
             body_params=np.load(f'{d}/body_params_opt_clip_0.npy'),
             contact_lbl=np.load(f'{d}/contact_lbl_rec_clip_0.npy'))
    # PROX index tables: OpenPose(118) <- SMPL-X(127) joint map (temp_prox/misc_utils.py:87-197 called as
    # data_parser_slide.py:225-229 does) and the friction vertex set L_Leg + R_Leg + gluteus
    # (fit_temp_loadprox_slide.py:349-354, again through list(set(...)) order)
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_misc_utils', f'{REF}/temp_prox/misc_utils.py')
    mu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mu)
    jmap = mu.smpl_to_openpose('smplx', use_hands=True, use_face=True, use_face_contour=False, openpose_format='coco25')
    fric = []
    for seg in ('L_Leg', 'R_Leg', 'gluteus'):
        with open(f'{REF}/body_segments/{seg}.json') as f:
            fric.append(list(set(json.load(f)['verts_ind'])))
    np.savez(os.path.join(OUT, 'prox_tables.npz'), joint_map=np.asarray(jmap, np.int32),
             contact_fric_verts_ids=np.concatenate(fric).astype(np.int32))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    sys.exit(main())
