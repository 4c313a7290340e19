#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "seeq_dfa.h"
static unsigned long long rs=88172645463325252ull;
static inline unsigned rnd(){ rs^=rs<<13; rs^=rs>>7; rs^=rs<<17; return (unsigned)(rs>>11); }
/* logical automaton from the physical stream table (undo rotation) */
int main(int argc,char**argv){
  const char*pat= argc>1?argv[1]:"GATGTAGCGCGATTAGCCTG"; int tau=argc>2?atoi(argv[2]):3; char keys[64]; int m=0; for(const char*c=pat;*c;c++) keys[m++]=*c=='A'?1:*c=='C'?2:*c=='G'?4:8;
  seeq_dfa_t*d=seeq_dfa_build_stream(keys,m,tau);
  int R=d->nrows; printf("rows %d\n",R);
  /* logical next[row][col] -> row */
  static int nx[5000][8];
  for(int r=0;r<R;r++) for(int k=0;k<8;k++){ unsigned sv=d->table[r*8 + (k ^ ((r&1)?4:0))]; nx[r][k]=sv>>4; }
  /* frequency by simulation */
  static unsigned long long freq[5000]; int s=0; int lp=0; const int colof[4]={0,1,2,3};
  for(long t=0;t<2000000;t++){ int col; if(lp==150){col=5;lp=0;} else {col=rnd()&3; lp++;} s=nx[s][col]; freq[s]++; }
  int idx[5000]; for(int i=0;i<R;i++) idx[i]=i;
  for(int i=0;i<R;i++) for(int j=i+1;j<R;j++) if(freq[idx[j]]>freq[idx[i]]){int t=idx[i];idx[i]=idx[j];idx[j]=t;}
  static int rowof[5000]; for(int i=0;i<R;i++) rowof[idx[i]]=i;
  for(int variant=0;variant<4;variant++){
    /* address of (state r, col k) under variant: 0: current (row=r, rot=r&1); 1: row=r, rot=(r>>3)&1; 2: ranked rows, rot=(row>>3)&1 ; 3: ranked rows, rot = (row>>2)&... 8-byte rows? */
    rs=12345;
    unsigned st[64]; memset(st,0,sizeof st); int linepos[64]; for(int i=0;i<64;i++) linepos[i]=rnd()%151;
    double cyc=0; long steps=0;
    for(int t=0;t<20000;t++){
      unsigned addr[64];
      for(int l=0;l<64;l++){ int col; if(linepos[l]==150){col=5;linepos[l]=0;} else {col=rnd()&3; linepos[l]++;}
        int r=st[l]; int row = variant>=2? rowof[r]:r; int rot = variant==0? (row&1) : ((row>>3)&1);
        addr[l]= row*16 + ((col*2) ^ (rot?8:0)); st[l]=nx[r][col]; }
      for(int h=0;h<2;h++){ int maxc=0; unsigned seen[32][32]; int cnt[32]; memset(cnt,0,sizeof cnt);
        for(int l=h*32;l<h*32+32;l++){ unsigned dw=addr[l]>>2; int bk=dw%32; int f=0; for(int k=0;k<cnt[bk];k++) if(seen[bk][k]==dw){f=1;break;} if(!f) seen[bk][cnt[bk]++]=dw; }
        for(int bk=0;bk<32;bk++) if(cnt[bk]>maxc) maxc=cnt[bk]; cyc+=maxc; }
      steps++;
    }
    printf("variant %d: %.2f cycles/gather\n", variant, cyc/steps);
  }
  return 0; }
