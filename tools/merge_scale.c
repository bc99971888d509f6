/* tools/merge_scale.c -- mi_memfs_update_from_entries + mi_memfs_add_layer_by_scan at C4 entry counts straight through the C ABI (no Python
 * in the process): merge_scale <directories of 100 files>.  gcc -O2 -I include tools/merge_scale.c -L makisu_amd -lmakisu_mi -Wl,-rpath,$PWD/makisu_amd
 * MI_MOUNTS_FILE=/dev/null keeps the box's own mount table out of it.  profiles/r05_host_scale.txt */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "makisu_mi.h"
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
int main(int argc,char**argv){
  long nd = argc>1?atol(argv[1]):100000, per=100;
  long n = nd*(per+1);
  mi_tree_entry* e = calloc(n,sizeof *e);
  char* names = malloc(n*40);
  long k=0;
  for(long d=0;d<nd;++d){
    char* p=names+k*40; sprintf(p,"dir%07ld",d);
    e[k].relpath=p; e[k].kind=0; e[k].mode=040755; e[k].mtime_sec=100; e[k].file_index=-1; ++k;
    for(long f=0;f<per;++f){ p=names+k*40; sprintf(p,"dir%07ld/file%04ld.bin",d,f);
      e[k].relpath=p; e[k].kind=1; e[k].mode=0100644; e[k].mtime_sec=100; e[k].size=4096; e[k].file_index=-1; ++k; }
  }
  mi_memfs* fs; if(mi_memfs_create("/tmp",NULL,0,0,&fs)) return 1;
  uint64_t merged=0; double t0=now();
  int rc=mi_memfs_update_from_entries(fs,e,n,&merged);
  double dt=now()-t0;
  printf("%ld entries: merge rc=%d merged=%llu in %.3f s = %.3f us/entry\n",n,rc,(unsigned long long)merged,dt,dt*1e6/n);
  t0=now();
  mi_copy_layer* l; uint64_t ne;
  for(long i=0;i<n;++i) if(e[i].kind==1) e[i].file_index=-1;
  rc=mi_memfs_add_layer_by_scan(fs,e,n,NULL,0,&l,&ne);
  dt=now()-t0;
  printf("scan rc=%d layer=%llu in %.3f s = %.3f us/entry\n",rc,(unsigned long long)ne,dt,dt*1e6/n);
  FILE* f=fopen("/proc/self/status","r"); char line[256]; while(fgets(line,sizeof line,f)) if(!strncmp(line,"VmHWM",5)) fputs(line,stdout);
  return 0;
}
