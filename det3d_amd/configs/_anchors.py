"""Helpers shared by the in-repo multi-group configs: expand a compact per-class table into the
`tasks` / `anchor_generators` lists the reference's config files spell out by hand."""


def expand(groups, table, xy_range, velocities=None):
    """groups: list of class-name lists (one per head); table: class -> ((w,l,h), z, matched, unmatched)."""
    tasks = [dict(num_class=len(g), class_names=list(g)) for g in groups]
    names = [c for g in groups for c in g]
    gens = []
    for c in names:
        size, z, hi, lo = table[c]
        g = dict(type="anchor_generator_range", class_name=c, sizes=list(size), rotations=[0, 1.57],
                 anchor_ranges=[-xy_range, -xy_range, z, xy_range, xy_range, z],
                 matched_threshold=hi, unmatched_threshold=lo)
        if velocities is not None:
            g["velocities"] = list(velocities)
        gens.append(g)
    return tasks, names, gens


def iou_assigner(tasks, gens):
    return dict(type="iou", tasks=tasks, anchor_generators=gens, sample_positive_fraction=-1, sample_size=512,
                pos_area_threshold=-1, region_similarity_calculator=dict(type="nearest_iou_similarity"))
