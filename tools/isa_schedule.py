"""Print the instruction order of a kernel's main loop from hipcc -S output: M = MFMA, D = LDS-DMA / buffer load,
r = ds_read, w = s_waitcnt, B = s_barrier, j = branch.  Usage: isa_schedule.py file.s <mangled-name-substring> ..."""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    for name in sys.argv[2:]:
        for m in re.finditer(r'^(_Z\w*' + re.escape(name) + r'\w*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
            lines = m.group(2).split('\n')
            best = None
            for st in [i for i, l in enumerate(lines) if 's_barrier' in l]:
                for i in range(st, len(lines)):
                    if 's_cbranch' in lines[i] and i > st + 40:
                        seg = lines[st:i + 1]
                        n = sum(1 for l in seg if 'v_mfma' in l)
                        if best is None or n > best[0]:
                            best = (n, seg)
                        break
            ev = []
            for l in best[1]:
                t = l.strip()
                for pre, ch in (('v_mfma', 'M'), ('buffer_load', 'D'), ('ds_read', 'r'), ('ds_load', 'r'), ('s_waitcnt', 'w'),
                                ('s_barrier', 'B'), ('s_cbranch', 'j')):
                    if t.startswith(pre):
                        ev.append(ch)
            vg = re.search(re.escape(m.group(1)) + r'.*?\.amdhsa_next_free_vgpr (\d+)', s, re.S)
            print(m.group(1), 'vgpr', vg.group(1) if vg else '?')
            print('  ', ''.join(ev))


if __name__ == '__main__':
    main()
