"""CPU model of the integrator chain's momentum exchange (openmmtools_amd/csrc/integrate.hip, token 'M'): G workgroups of one replica,
each with a partial sum per epoch, exchange them through 64-bit words [parity][workgroup] that carry the epoch's low 16 bits as a tag
and a 48-bit payload.  A workgroup publishes its word (a store that becomes visible some time later) and then reads every word of
its replica, spinning on a word until its tag is the epoch's; then it goes on to the next epoch.  The model runs the workgroups
under an arbitrary interleaving with arbitrary store delays and checks that every workgroup obtains the true sum of every epoch --
in particular that a word of epoch e - 2 (same parity half) or e - 65536 (same tag) can never be taken for epoch e's.

The argument the kernel's comment makes, made executable: a workgroup rewrites its word of parity p two epochs later, and it can only
get there after finishing the epoch in between, which needs every other workgroup's word of that epoch, which those publish only
after they are done reading epoch e.

usage: python tools/experiments/chain_exchange_model.py [n_schedules]"""
import random
import sys

MASK48 = (1 << 48) - 1


def pack(value, epoch):
    return ((value & MASK48) << 16) | (epoch & 0xffff)


def unpack(word):
    v = word >> 16
    return v - (1 << 48) if v >> 47 else v


def run(G, n_epochs, seed, first_epoch=1, max_delay=6):
    """returns the number of scheduler steps; raises AssertionError on a wrong sum"""
    rng = random.Random(seed)
    mem = [[0] * G for _ in range(2)]                       # visible words (memset 0 by the host)
    in_flight = []                                          # (due_step, parity, g, word): stores on their way
    part = [[rng.randrange(-(1 << 40), 1 << 40) for _ in range(G)] for _ in range(n_epochs)]
    truth = [sum(p) for p in part]
    # per workgroup: epoch index, phase (0 publish, 1 read), next word to read, running sum
    st = [dict(e=0, phase=0, q=0, acc=0) for _ in range(G)]
    done = 0
    step = 0
    while done < G:
        step += 1
        assert step < 10_000_000, 'no progress: deadlock'
        # stores land (a workgroup's own stores stay in program order: same parity slot is only rewritten two epochs later)
        still = []
        for item in in_flight:
            if item[0] <= step:
                mem[item[1]][item[2]] = item[3]
            else:
                still.append(item)
        in_flight = still
        g = rng.randrange(G)
        s = st[g]
        if s['e'] >= n_epochs:
            continue
        epoch = first_epoch + s['e']
        par = epoch & 1
        if s['phase'] == 0:
            in_flight.append((step + rng.randrange(1, max_delay + 1), par, g, pack(part[s['e']][g], epoch)))
            s['phase'], s['q'], s['acc'] = 1, 0, 0
        else:
            w = mem[par][s['q']]
            if (w & 0xffff) == (epoch & 0xffff):             # the spin: otherwise try again at a later step
                s['acc'] += unpack(w)
                s['q'] += 1
                if s['q'] == G:
                    assert s['acc'] == truth[s['e']], ('wrong sum', g, epoch, s['acc'], truth[s['e']])
                    s['e'] += 1
                    s['phase'] = 0
                    if s['e'] == n_epochs:
                        done += 1
    return step


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for k in range(n):
        G = random.Random(k).choice([1, 2, 3, 5, 32])
        run(G, 40, seed=k)
        run(G, 12, seed=k + 10_000, first_epoch=65530)        # the 16-bit tag wraps inside the run
    print('%d schedules x 2: every workgroup saw the true sum of every epoch' % n)
